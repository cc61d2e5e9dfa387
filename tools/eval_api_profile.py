"""Developer probe (GPU): cProfile of `get_Jacobian()(Grid)` + `compute_curl(X=Grid)` on the 64^3 grid (cold evaluator cache)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np
import torch

import spateo_amd as st
from spateo_amd import vectorfield as vfm

dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
rng = np.random.default_rng(0)
ctrl = rng.uniform(-200, 200, (500, 3))
grid = np.stack([g.ravel() for g in np.meshgrid(*[np.linspace(-250, 250, 64)] * 3, indexing="ij")], axis=1)
vf = st.SvcVectorField(dtype=dtype, device="cuda:0")
vf.vf_dict = {"X_ctrl": ctrl, "C": rng.standard_normal((500, 3)), "beta": 1e-4}


def pair():
    vfm.clear_eval_cache()
    J = vf.get_Jacobian()(grid)
    c = vf.compute_curl(X=grid)
    return J, c


for _ in range(3):
    pair()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pair()
    ts.append(1e3 * (time.perf_counter() - t0))
print("wall ms per pair:", [round(t, 2) for t in ts])
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    pair()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
