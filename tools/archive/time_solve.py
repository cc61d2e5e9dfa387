"""Developer probe (GPU): time mvf_solve at several sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd._kernels import HipKernels
k = HipKernels("cuda:0", "float64")
for m in [int(a) for a in sys.argv[1:]] or [500, 2000, 3000]:
    rng = np.random.default_rng(0); A = rng.standard_normal((m, m + 50)); G = torch.from_numpy(A @ A.T / m).cuda()
    K = torch.eye(m, dtype=torch.float64, device="cuda"); R = torch.randn(m, 3, dtype=torch.float64, device="cuda")
    C = torch.empty(m, 3, dtype=torch.float64, device="cuda"); info = torch.zeros(1, dtype=torch.int32, device="cuda")
    k.solve(G, K, 0.1, 0.0, R, C, info); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): k.solve(G, K, 0.1, 0.0, R, C, info)
    e1.record(); torch.cuda.synchronize()
    Cr = torch.linalg.solve(G + 0.1 * K, R)
    print(m, "solve ms", e0.elapsed_time(e1) / 5, "err", float((C - Cr).abs().max() / Cr.abs().max()), int(info))
