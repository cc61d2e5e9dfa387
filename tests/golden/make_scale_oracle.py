#!/usr/bin/env python
"""Writes tests/golden/scale_oracle.npz: the float64 ORACLE's outputs (this repo's restatement, not the reference - the
reference cannot run: dynamo is absent) for the seeded M = 2000 / 3000 cases of tests/test_gpu_scale.py, so that the GPU
suite does not spend minutes of host time in 3000 x 3000 lstsq calls.  It calls the very compute functions of the tests
(single EM step and 10-step fits, lambda_ = 3 and 0.02, plus the lstsq-vs-eigh and float32-kernel floors).

    python tests/golden/make_scale_oracle.py        (about 10 minutes on 8 cores)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), os.path.dirname(HERE)):
    sys.path.insert(0, p)
os.environ["MVF_SCALE_ORACLE_LIVE"] = "1"

import test_gpu_scale as T  # noqa: E402


def main():
    for M in (2000, 3000):
        for lam in (3.0, 0.02):
            T._large_m_case(M, lam)
            print("fit", M, lam, "done", flush=True)
    for M in (2000, 3000):
        for lam in (3.0, 0.02):
            T._single_step_case(M, lam)
            print("step", M, lam, "done", flush=True)
    out = {f"{k}|{f}": v for k, d in T._ORACLE_STORE.items() for f, v in d.items()}
    path = os.path.join(HERE, "scale_oracle.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
