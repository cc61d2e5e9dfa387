"""``spateo_amd.tdr``: the morphometric vector-field slice of ``spateo.tdr`` (``spateo/tdr/__init__.py:1-9``)."""
from .interpolations import get_X_Y_grid, kernel_interpolation
from .morphometrics import *  # noqa: F401,F403
from .morphometrics import _morphofield_sparsevfc  # noqa: F401
