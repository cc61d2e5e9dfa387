import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/spateo-release_amd")
import numpy as np, torch
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess
for name, N, M in (("C5 organ", 250_000, 500), ("C2", 50_000, 500)):
    X, V, _ = make_config("C2", N=N, seed=101)
    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0, device="cuda:0")
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float32", device="cuda:0")
    eng.init_state(gamma=0.9)
    ts = []
    for it in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        E, tecr = eng.em_step(a=5, lambda_=0.02, minP=1e-5, theta=0.75)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    st = eng.solver_stats
    print(name, "ms/iter", [round(t, 2) for t in ts])
    print("   rank", st.get("rank"), "factor_rank", st.get("factor_rank"), "block", st.get("block"), "sweeps", st.get("sweeps"), "async", st.get("async"), "cholesky", st.get("cholesky"))
