import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/spateo-release_amd")
import numpy as np, torch
from spateo_amd import _lib
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess
X, V, _ = make_config("C2", N=250_000, seed=101)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=500, seed=0, device="cuda:0")
_lib.debug_option("lr_timing", 1)
eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float32", device="cuda:0")
eng.async_direct = False
eng.init_state(gamma=0.9)
for it in range(8):
    sys.stderr.write(f"--- iteration {it + 1}\n"); sys.stderr.flush()
    eng.em_step(a=5, lambda_=0.02, minP=1e-5, theta=0.75)
    torch.cuda.synchronize()
