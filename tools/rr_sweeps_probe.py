"""The one-launch 64 x 64 Rayleigh-Ritz of the direct form (M <= 640) in the EM's steady state: active rounds / rotations per
inner sweep (developer option lr_timing, synchronous entry) for every EM iteration of a C2-sized fit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import torch
from spateo_amd import _lib
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess
n, iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000, int(sys.argv[2]) if len(sys.argv) > 2 else 24
X, V, _ = make_config("C2", N=n)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=500, seed=0, device="cuda:0")
_lib.debug_option("lr_timing", 1)
eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float32", device="cuda:0")
eng.async_direct = False
eng.init_state(gamma=0.9)
for it in range(iters):
    sys.stderr.write(f"--- iteration {it + 1} sigma2 {eng.sigma2:.6e}\n"); sys.stderr.flush()
    eng.em_step(a=5, lambda_=0.02, minP=1e-5, theta=0.75)
    torch.cuda.synchronize()
