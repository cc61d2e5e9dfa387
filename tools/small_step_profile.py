#!/usr/bin/env python
"""Developer tool: run EM steps of a small configuration (default C2: 50k cells x 500 control points) so that
`rocprofv3 --kernel-trace --stats` shows where a launch-bound step spends its time (kernel time vs wall time)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "spateo-release_amd")]
import torch
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dtype = sys.argv[3] if len(sys.argv) > 3 else "float32"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
X, V, _ = make_config("C2", N=N)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0")
eng.init_state(0.9)
kw = dict(a=5.0, lambda_=0.02, minP=1e-5, theta=0.75)
for _ in range(5):
    eng.em_step(**kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.em_step(**kw)
torch.cuda.synchronize()
print(f"N={N} M={M} {dtype}: {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step wall over {steps} steps, jitter={eng.jitter:g}")
