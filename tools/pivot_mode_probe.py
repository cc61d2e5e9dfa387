"""GPU probe of gram_mode="pivot:K": deviation from the oracle fixtures per quantity (in units of the reference's own floor)
for K = 1 .. 4 rank-revealing iterations before the switch, both cell dtypes.  python tools/pivot_mode_probe.py [cases]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import spateo_amd as st  # noqa: E402
import test_gpu_scale as T  # noqa: E402

cases = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c4_200k", "m3000_20k", "m2000_20k"]
Ks = sys.argv[2].split(",") if len(sys.argv) > 2 else "1,2,3,4".split(",")
out = []
for case in cases:
    if case == "c4_200k":
        X, V, kw, ref, table = T._c4_sample_case()
        stride = T._C4_SAMPLE["stride"]
    elif case in ("c4_rank", "c3_full"):
        from spateo_amd._synthetic import make_config
        fx, ref, table = T._stream_fixture(case)
        cfg, n, M = T._STREAM_CASES[case]
        X, V, _ = make_config(cfg, N=n)
        kw = dict(M=M, lambda_=0.02, lstsq_method="scipy", MaxIter=int(fx["steps"]), ecr=0.0, seed=0)
        stride = int(fx["stride"])
    else:
        X, V, kw, ref, table = T._large_m_case(3000 if case == "m3000_20k" else 2000, 0.02)
        stride = 1
    for mode in ["full"] + [f"pivot:{k}" for k in Ks]:
        for dtype in ("float64", "float32"):
            got = st.SparseVFC(X, V, None, dtype=dtype, device="cuda:0", gram_mode=mode, **kw)
            dev = T._fixture_devs(got, ref, stride)
            fl = {q: table[q][0] for q in dev}
            rec = {"case": case, "mode": mode, "dtype": dtype, "ctrl_used": int(len(got.get("ctrl_subset", range(kw["M"])))),
                   **{q: dev[q] for q in dev}, **{q + "_over_floor": dev[q] / fl[q] for q in dev}}
            out.append(rec)
            print(json.dumps(rec), flush=True)
            del got
            torch.cuda.empty_cache()
