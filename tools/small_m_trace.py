"""Developer probe (GPU, run under rocprofv3 --kernel-trace): 8 EM steps of C2 (50 k x 500, float32) with the deflated solve."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import torch

from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

method = sys.argv[1] if len(sys.argv) > 1 else "deflated"
X, V, M = make_config("C2")
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
SparseVFCEngine.minnorm_method = method
eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float32", device="cuda:0")
eng.init_state(0.9)
for _ in range(8):
    eng.em_step(a=5.0, lambda_=0.02, minP=1e-5, theta=0.75)
torch.cuda.synchronize()
print({k: (v[-3:] if isinstance(v, list) else v) for k, v in eng.solver_stats.items()})
