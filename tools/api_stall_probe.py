"""Every public entry point once more on medium-sized data with the caller KEEPING what it gets, under a tracer that reports any single
tensor transfer / staging operation above 3 ms: is a pageable transfer or a many-threaded CPU op left anywhere on the API paths?"""
import os, sys, time, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "spateo-release_amd")]
import numpy as np, torch
import spateo_amd as st
from spateo_amd._synthetic import make_config

slow = []
def wrap(name, f):
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
        if dt > 3e-3:
            fr = traceback.extract_stack(limit=5)[:-1]
            sh = tuple(a[0].shape) if a and hasattr(a[0], "shape") else ""
            slow.append(f"{name} {1e3 * dt:.1f} ms {sh} <- " + " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in reversed(fr)))
        return r
    return g
torch.Tensor.to = wrap("Tensor.to", torch.Tensor.to); torch.Tensor.cpu = wrap("Tensor.cpu", torch.Tensor.cpu)
torch.Tensor.copy_ = wrap("Tensor.copy_", torch.Tensor.copy_)

n = 60_000
X, V, _ = make_config("C2", N=n)
keep = []
def timed(tag, fn, reps=3):
    ts = []
    for _ in range(reps):
        slow.clear(); t0 = time.perf_counter(); keep.append(fn()); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{tag}: {[round(t, 1) for t in ts]} ms" + ("" if not slow else "   SLOW OPS (last call): " + " | ".join(slow)))

ad = st.AnnDataLite(obsm={"align_spatial": X.copy(), "V_mapping": V.copy()})
timed("morphofield_sparsevfc (M = 100, restart loop)", lambda: st.tdr.morphofield_sparsevfc(ad, NX=X[::50].copy(), M=100, dtype="float32"), reps=2)
for nm in ("velocity", "jacobian", "divergence", "curl", "acceleration", "curvature", "torsion"):
    timed(f"morphofield_{nm}", lambda nm=nm: getattr(st.tdr, f"morphofield_{nm}")(ad))
timed("SparseVFC M = 500 float64 + grid", lambda: st.SparseVFC(X, V, X[::40] + 1.0, M=500, lambda_=0.02, dtype="float64"))
vf = keep[-1]
timed("vector_field_function (60 k points)", lambda: st.vector_field_function(X + 0.5, vf))
timed("con_K 60 k x 500", lambda: st.con_K(X, vf["X_ctrl"], vf["beta"]))
G = np.random.default_rng(0).standard_normal((n, 16))
src = st.AnnDataLite(obsm={"spatial": X.copy()}, X=G, var_names=[f"g{i}" for i in range(16)])
timed("kernel_interpolation Dy = 16", lambda: st.tdr.kernel_interpolation(src, target_points=X[::20].copy(), keys=[f"g{i}" for i in range(16)], M=300, dtype="float32"), reps=2)
a2 = st.AnnDataLite(obsm={"align_spatial": X[:2000].copy()})
a2.uns["VecFld_morpho"] = dict(ad.uns["VecFld_morpho"])
a2.uns["VecFld_morpho"]["X"] = X[:2000].copy()
a2.uns["VecFld_morpho"]["V"] = np.asarray(ad.uns["VecFld_morpho"]["V"])[:2000].copy()
timed("morphopath (2000 cells, 100 points)", lambda: st.tdr.morphopath(a2, t_end=50.0, interpolation_num=100), reps=2)
print("cgroup:", open("/sys/fs/cgroup/cpu.stat").read().split("nr_throttled")[1].split()[0] if os.path.exists("/sys/fs/cgroup/cpu.stat") else "n/a", "throttled periods")
