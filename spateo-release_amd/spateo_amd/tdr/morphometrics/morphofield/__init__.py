from .gaussian_process import morphofield_gp
from .sparsevfc import _morphofield_sparsevfc, cell_directions, morphofield_sparsevfc
from .trajectory import construct_genesis_states, morphopath
