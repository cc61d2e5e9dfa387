from .interpolation_sparseVFC import kernel_interpolation
from .utils import get_X_Y_grid
