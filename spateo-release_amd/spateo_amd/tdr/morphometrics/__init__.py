from .morphofield import (
    _morphofield_sparsevfc,
    cell_directions,
    construct_genesis_states,
    morphofield_gp,
    morphofield_sparsevfc,
    morphopath,
)
from .morphofield_dg import (
    morphofield_acceleration,
    morphofield_curl,
    morphofield_curvature,
    morphofield_divergence,
    morphofield_jacobian,
    morphofield_torsion,
    morphofield_velocity,
)
