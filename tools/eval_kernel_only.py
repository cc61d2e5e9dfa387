"""The evaluator kernel alone (64^3 grid x 500 control points, all eight quantities), 20 launches - what the PMC / kernel-trace
passes of tools/gpu_r6_o.sh run under rocprofv3.   python tools/eval_kernel_only.py [float32|float64]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spateo_amd import vectorfield as vfm  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
rng = np.random.default_rng(0)
M = 500
ctrl = rng.uniform(-1, 1, (M, 3)) * np.array([200.0, 120.0, 90.0])
g = [np.linspace(-a, a, 64) for a in (200.0, 120.0, 90.0)]
Grid = np.stack(np.meshgrid(*g, indexing="ij"), -1).reshape(-1, 3)
C = rng.standard_normal((M, 3))
beta = 1.0 / 40.0 ** 2
k = vfm._shared_kernels("cuda:0", dtype)
c = ctrl.mean(0)
x4, c4 = k.to_x4(Grid, c), k.to_x4(ctrl, c)
Cd = torch.from_numpy(C).to("cuda:0")
k.eval(x4, c4, beta, Cd, vfm._EVAL_ALL)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    k.eval(x4, c4, beta, Cd, vfm._EVAL_ALL)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{dtype}: {ms:.4f} ms per launch, {len(Grid) * M / ms / 1e6:.1f} Gpairs/s, "
      f"{len(Grid) * M * 24 / ms / 1e9:.2f} TF of the 78.6 TF float64 peak = {len(Grid) * M * 24 / ms / 1e9 / 78.6:.3f}")
