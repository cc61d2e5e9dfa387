"""Developer probe (GPU): `integrate_kernel` at BASELINE config 2's shape - 50 k trajectories x 250 output samples against
M = 500 control points (morphopath on the C2 field).  Times the fused RK4 launch with HIP events on its stream in both
sampling plans of `integrate_field` (uniform_time: 250 samples x 4 substeps; arc_length: 1001 dense samples x 2 substeps)
and prints pairs/s and the fraction of the 78.6 TF vector-f64 peak at the per-pair work of the field evaluation
(v only: 3 subtractions + 3 FMA squared distance + exp2 + 3 f64 FMA accumulation ~ 14 flop + exp2 per pair).
Run under `rocprofv3 --kernel-trace --stats` for the committed kernel row (profiles/r05_integrate_kernel_stats.md)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np
import torch

import spateo_amd as st
from spateo_amd._kernels import HipKernels
from spateo_amd._synthetic import make_config

X, V, M = make_config("C2")
vf = st.SparseVFC(X, V, None, M=M, lambda_=0.02, lstsq_method="scipy", dtype="float32", device="cuda:0", MaxIter=30)
n = len(X)
t_end = 50.0
out = {"trajectories": n, "ctrl": M, "samples": 250}
for dtype in ("float32", "float64"):
    k = HipKernels("cuda:0", dtype)
    c = vf["X_ctrl"].mean(0)
    x4, c4 = k.to_x4(X, c), k.to_x4(vf["X_ctrl"], c)
    Cd = torch.from_numpy(np.ascontiguousarray(vf["C"])).cuda()
    for plan, n_out, sub in (("uniform_time", 250, 4), ("arc_length", 1001, 2)):
        dt = t_end / (n_out - 1)
        tr = k.integrate(x4, c4, vf["beta"], Cd, dt, sub, n_out)
        torch.cuda.synchronize()
        del tr
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 3
        for _ in range(reps):
            tr = k.integrate(x4, c4, vf["beta"], Cd, dt, sub, n_out)
            del tr
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        pairs = float(n) * (n_out - 1) * sub * 4 * M  # RK4: four field evaluations per step
        rate = pairs / (ms * 1e-3)
        out[f"{dtype}|{plan}"] = {"ms": ms, "rk4_steps": (n_out - 1) * sub, "pairs": pairs, "Tpairs_per_s": rate / 1e12,
                                 "f64_valu_TFLOPs_at_14_flop_per_pair": rate * 14 / 1e12,
                                 "frac_of_78.6_TF": rate * 14 / 1e12 / 78.6,
                                 "traj_output_GB": n * n_out * 3 * 8 / 1e9}
        print(f"integrate[{dtype}] {plan}: {ms:.1f} ms, {rate / 1e12:.3f} T pairs/s", flush=True)
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "r05_integrate_probe.json"), "w") as fh:
    json.dump(out, fh, indent=1)
