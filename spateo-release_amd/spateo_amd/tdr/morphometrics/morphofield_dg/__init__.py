from .differential_geometry import (
    morphofield_acceleration,
    morphofield_curl,
    morphofield_curvature,
    morphofield_divergence,
    morphofield_jacobian,
    morphofield_torsion,
    morphofield_velocity,
)
