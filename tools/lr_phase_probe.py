"""Developer probe (GPU): phase times of the deflated coefficient solve inside the EM loop (developer option lr_timing).

    python tools/lr_phase_probe.py M N [steps]      -> the library's own per-phase lines (stderr) for the last three steps
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import torch

from spateo_amd import _lib
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

M, N = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
X, Y, _ = make_config("C2", N=N)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, Y, M=M, seed=0)
eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float64", device="cuda:0")
eng.init_state()
for i in range(steps):
    if i == steps - 3:
        _lib.debug_option("lr_timing", 1)
    eng.em_step(lambda_=0.02)
torch.cuda.synchronize()
_lib.debug_option("lr_timing", 0)
print(eng.solver_stats["rank"][-3:], eng.solver_stats.get("block", [])[-3:], eng.solver_stats["sweeps"][-3:])
