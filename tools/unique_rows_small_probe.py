"""Device vs host `unique_rows` below the 200 k-row threshold: where should the device path start?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "spateo-release_amd")]
import numpy as np, torch
from spateo_amd._kernels import HipKernels
from spateo_amd._synthetic import make_config
import spateo_amd.preprocess as pre
k = HipKernels("cuda:0", "float64")
for n in (10_000, 20_000, 50_000, 100_000, 200_000):
    X, _, _ = make_config("C2", N=n)
    X[5] = X[7]
    td, th = [], []
    for rep in range(7):
        torch.cuda.synchronize(); t = time.perf_counter(); S, idx = k.unique_rows(X); torch.cuda.synchronize(); td.append(time.perf_counter() - t)
    old = pre._DEVICE_UNIQUE_MIN_ROWS; pre._DEVICE_UNIQUE_MIN_ROWS = 10**12
    for rep in range(7):
        t = time.perf_counter(); Sh, ih = pre.unique_rows(X); th.append(time.perf_counter() - t)
    pre._DEVICE_UNIQUE_MIN_ROWS = old
    print(f"n = {n}: device {1e3 * min(td):.2f} ms, host {1e3 * min(th):.2f} ms, identical {np.array_equal(S, Sh) and np.array_equal(idx, ih)}")
