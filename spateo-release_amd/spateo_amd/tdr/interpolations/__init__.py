from .utils import get_X_Y_grid
