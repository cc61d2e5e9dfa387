#!/usr/bin/env python
"""Writes tests/golden/scale_oracle.npz: the float64 ORACLE's outputs (this repo's restatement, not the reference - the
reference cannot run: dynamo is absent) for the seeded cases of tests/test_gpu_scale.py that need minutes of host time:
M = 2000 / 3000 at 20 k cells (single EM step and 10-step fits, lambda_ = 3 and 0.02) and the bench's own generator at
BASELINE.md section 3's N_cpu (C4, 200 k cells x 3000, lambda_ = 0.02, 10 steps; every 8th cell of V / P stored).  Each
case carries the reference noise floors of tests/_floors.py (LAPACK driver swapped, Gram summation order changed,
float32 kernel values), computed on all cells.  It calls the very compute functions of the tests.

    python tests/golden/make_scale_oracle.py        (about 50 minutes on 8 cores, 12 GB of RAM)
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


def save():
    out = {f"{k}|{f}": v for k, d in T._ORACLE_STORE.items() for f, v in d.items()}
    path = os.path.join(HERE, os.environ.get("MVF_SCALE_ORACLE_OUT", "scale_oracle.npz"))
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path) / 1e6, "MB", flush=True)


def main():
    for M in (2000, 3000):
        for lam in (3.0, 0.02):
            T._single_step_case(M, lam)
            print("step", M, lam, "done", flush=True)
    save()
    for M in (2000, 3000):
        for lam in (3.0, 0.02):
            T._large_m_case(M, lam)
            print("fit", M, lam, "done", flush=True)
            save()
    T._c4_sample_case()
    print("C4 sample done", flush=True)
    save()


if __name__ == "__main__":
    main()
