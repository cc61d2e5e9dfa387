"""GPU probe of the evaluator path AT THE API (VERDICT r3 "next" #4): BASELINE config 2's "Jacobian + curl on the 64^3
grid" as the reference calls it - ``get_Jacobian()(Grid)`` then ``compute_curl(X=Grid)``, host arrays in, host arrays
out - wall time per call pair with a cold evaluator cache, the kernel time inside it (HIP events), and
``morphofield_jacobian``'s Jacobian + determinant on N cells.  One JSON line.
python tools/eval_api_probe.py [n_big_cells]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import spateo_amd as st  # noqa: E402
from spateo_amd import _lib, vectorfield as vfm  # noqa: E402
from spateo_amd._synthetic import make_config  # noqa: E402
from spateo_amd.tdr.interpolations.utils import get_X_Y_grid  # noqa: E402

n_big = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
X, V, M = make_config("C2")
_, _, Grid, _ = get_X_Y_grid(X=X, Y=V, grid_num=[64, 64, 64])
fit = st.SparseVFC(X, V, None, M=M, lambda_=0.02, lstsq_method="scipy", dtype="float32", device="cuda:0", MaxIter=30)
out = {"grid_points": len(Grid), "M": M}
for dtype in ("float64", "float32"):
    vf = st.SvcVectorField(dtype=dtype, device="cuda:0")
    vf.vf_dict = fit
    walls, kern = [], []
    k = vfm._shared_kernels("cuda:0", dtype)
    real = k.eval

    def timed(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real(*a, **kw)
        e1.record()
        kern.append((e0, e1))
        return r

    k.eval = timed
    for rep in range(7):
        vfm.clear_eval_cache()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        J = vf.get_Jacobian()(Grid)
        t1 = time.perf_counter()
        curl = vf.compute_curl(X=Grid)
        t2 = time.perf_counter()
        walls.append((1e3 * (t2 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    k.eval = real
    torch.cuda.synchronize()
    kms = [a.elapsed_time(b) for a, b in kern]
    w = np.array(walls[2:])
    pairs = len(Grid) * M
    out[dtype] = {"jacobian_plus_curl_wall_ms_median": float(np.median(w[:, 0])), "jacobian_call_ms": float(np.median(w[:, 1])),
                  "curl_call_ms": float(np.median(w[:, 2])), "first_call_pair_ms": walls[0][0],
                  "eval_launches_per_pair": len(kms) / len(walls), "eval_kernel_ms_all_quantities": float(np.median(kms[2:])),
                  "Gpairs_per_s": pairs / np.median(kms[2:]) / 1e6, "J_shape": list(J.shape), "curl_shape": list(curl.shape)}
    # the same two quantities through the kernel with only their flags (no other outputs written): kernel time alone
    c = fit["X_ctrl"].mean(0)
    x4, c4 = k.to_x4(Grid, c), k.to_x4(fit["X_ctrl"], c)
    Cd = torch.from_numpy(np.ascontiguousarray(fit["C"])).to("cuda:0")
    for name, fl in (("jac_curl_only", _lib.EVAL_JAC | _lib.EVAL_CURL), ("all", vfm._EVAL_ALL)):
        real(x4, c4, fit["beta"], Cd, fl)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            real(x4, c4, fit["beta"], Cd, fl)
        e1.record()
        torch.cuda.synchronize()
        out[dtype][f"kernel_ms_{name}"] = e0.elapsed_time(e1) / 20
# morphofield_jacobian's work at scale: Jacobians + determinants of n_big cells (host in, host out), float32 kernel values
Xb, _, _ = make_config("C3", N=n_big)
Xb = Xb * (np.abs(X).max() / np.abs(Xb).max())      # inside the C2 field's support
vf = st.SvcVectorField(dtype="float32", device="cuda:0")
vf.vf_dict = fit
for rep in range(2):
    vfm.clear_eval_cache()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Js, det = vf.jacobian_with_det(Xb)
    t1 = time.perf_counter()
t2 = time.perf_counter()
det_host = np.linalg.det(np.moveaxis(Js[:, :, :200_000], 2, 0))
t3 = time.perf_counter()
out["jacobian_with_det"] = {"cells": n_big, "wall_ms": 1e3 * (t1 - t0), "host_det_ms_per_200k_cells": 1e3 * (t3 - t2),
                            "det_vs_host_maxabs": float(np.abs(det[:200_000] - det_host).max()),
                            "det_scale": float(np.abs(det_host).max())}
print(json.dumps(out))
