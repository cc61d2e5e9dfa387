"""Developer probe (GPU): 40 EM iterations of C2 (50 k x 500) with the deflated solve, phase timing of every solve on stderr."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import torch

from spateo_amd import _lib
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
X, V, M = make_config("C2")
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0")
eng.init_state(0.9)
_lib.debug_option("lr_timing", 1)
for i in range(40):
    eng.em_step(a=5.0, lambda_=0.02, minP=1e-5, theta=0.75)
    print(f"step {i}: sigma2 {eng.sigma2:.6g} rank {eng.solver_stats['rank'][-1:]} factor {eng.solver_stats.get('factor_rank', [None])[-1:]} "
          f"block {eng.solver_stats.get('block', [None])[-1:]}", file=sys.stderr, flush=True)
torch.cuda.synchronize()
