"""Developer probe (GPU): the EM step with a wide Y (kernel_interpolation: Dy genes) at BASELINE config 3's size, 2 M cells x
2000 control points - Dy = 3 against Dy = 48 through the MFMA kernels on the cached U (mvf_wide.hip) and through the
three-columns-at-a-time VALU path of rounds 1 - 5.

    python tools/wide_y_probe.py [N] [M] [Dy] [--out file.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "spateo-release_amd")):
    sys.path.insert(0, _p)
import torch

from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 2_000_000
M = int(args[1]) if len(args) > 1 else 2000
DY = int(args[2]) if len(args) > 2 else 48
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
X, V, _ = make_config("C3", N=N)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
rng = np.random.default_rng(0)
W = rng.standard_normal((3, DY)) / np.sqrt(3.0)
Ywide = Yv @ W + 0.05 * rng.standard_normal((len(Yv), DY))   # DY "genes": mixtures of the displacement components + noise


def run(Y, wide, steps=6, warm=2):
    SparseVFCEngine.wide_y = wide
    try:
        eng = SparseVFCEngine(Xv, Y, ctrl, beta, dtype="float32", device="cuda:0")
    finally:
        SparseVFCEngine.wide_y = True
    eng.init_state(0.9)
    kw = dict(a=5.0, lambda_=0.02, minP=1e-5, theta=0.75)
    for _ in range(warm):
        eng.em_step(**kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.em_step(**kw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    Vg = eng.results()[0]
    rec = dict(dy=Y.shape[1], wide=bool(eng.wide), ms_per_em_step=ms, sigma2=eng.sigma2)
    eng.k.drop_ublk()
    del eng
    torch.cuda.empty_cache()
    return rec, Vg


res = {"cells": int(len(Xv)), "ctrl": int(M)}
res["dy3"], _ = run(Yv, True)
res["wide"], Vw = run(Ywide, True)
res["narrow"], Vn = run(Ywide, False)
res["wide_over_dy3"] = res["wide"]["ms_per_em_step"] / res["dy3"]["ms_per_em_step"]
res["narrow_over_dy3"] = res["narrow"]["ms_per_em_step"] / res["dy3"]["ms_per_em_step"]
res["field_maxrel_wide_vs_narrow"] = float(np.abs(Vw - Vn).max() / np.abs(Vn).max())
print(json.dumps(res))
if out_path:
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
