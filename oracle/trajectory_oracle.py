"""CPU ORACLE (test infrastructure, NOT product code) for trajectory integration of the learned field.

``spateo.tdr.morphopath`` (``spateo/tdr/morphometrics/morphofield/trajectory.py:11-117``) delegates to dynamo's
``fate`` (adaptive RK45 + arc-length resampling), which is outside the reference tree and not restated here:
**parity unpinned**.  This oracle checks the thing that is well defined - the solution of dx/dt = v(x) at given times -
with SciPy's DOP853 at tight tolerances on the float64 oracle field."""
import numpy as np
from scipy.integrate import solve_ivp

from .sparsevfc_oracle import vector_field_function


def integrate(vf_dict, x0, t_eval, field=None):
    """Positions at ``t_eval`` (increasing from 0, or decreasing from 0) for each start point: (n, len(t_eval), d)."""
    f = field if field is not None else (lambda x: vector_field_function(x, vf_dict))
    out = []
    for p in np.atleast_2d(x0):
        sol = solve_ivp(lambda t, y: np.asarray(f(y)).reshape(-1), (t_eval[0], t_eval[-1]), p, t_eval=t_eval,
                        method="DOP853", rtol=1e-11, atol=1e-13)
        out.append(sol.y.T)
    return np.asarray(out)
