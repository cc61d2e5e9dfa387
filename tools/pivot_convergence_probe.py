"""Fits run to convergence (dynamo's stopping rule, ecr = 1e-5) in the default and the pivot mode: iterations, wall, final field."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
import spateo_amd as st
from spateo_amd._synthetic import make_config
for cfg, n, M in (("C4", 200_000, 3000), ("C4", 1_000_000, 3000), ("C3", 2_000_000, 2000)):
    X, V, _ = make_config(cfg, N=n)
    out = {}
    for mode in ("full", "pivot"):
        for rep in range(2):
            t0 = time.perf_counter()
            r = st.SparseVFC(X, V, None, M=M, lambda_=0.02, lstsq_method="scipy", seed=0, dtype="float32", device="cuda:0", gram_mode=mode)
            torch.cuda.synchronize(); wall = time.perf_counter() - t0
        out[mode] = r
        print(json.dumps({"case": f"{cfg} {n} x {M}", "mode": mode, "iterations": int(r["iteration"]) + 1, "wall_s": wall, "sigma2": r["sigma2"],
                          "ctrl_used": int(len(r.get("ctrl_subset", range(M)))), "E_last": float(r["E_traj"][-1]), "tecr_last": float(r["tecr_traj"][-1])}), flush=True)
    a, b = out["full"], out["pivot"]
    print(json.dumps({"case": f"{cfg} {n} x {M}", "pivot_vs_full_V": float(np.abs(a["V"] - b["V"]).max() / np.abs(a["V"]).max()),
                      "sigma2_rel": abs(a["sigma2"] - b["sigma2"]) / a["sigma2"], "P_max": float(np.abs(a["P"] - b["P"]).max())}), flush=True)
