"""Developer probe (GPU): materialised con_K bandwidth, the three store patterns of csrc/mvf_conk.hip (MVF_CONK = rows | flat | 2d), float32 and float64, at BASELINE config 3's shape (2 M x 2000) and at 1 M x 3000."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd import _lib as L
from spateo_amd._synthetic import make_config
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
out = {}
for nk, mk in ((2_000_000, 2000), (1_000_000, 3000)):
    X, V, _ = make_config("C4", N=nk)
    ctrl = X[np.random.default_rng(0).choice(nk, mk, replace=False)]
    for dt, code, tdt in (("float32", 0, torch.float32), ("float64", 1, torch.float64)):
        npdt = np.float32 if dt == "float32" else np.float64
        xs = torch.from_numpy((X - ctrl.mean(0)).astype(npdt)).cuda()
        cs = torch.from_numpy((ctrl - ctrl.mean(0)).astype(npdt)).cuda()
        K = torch.empty(nk, mk, dtype=tdt, device="cuda")
        for form in ("rows", "rows8", "flat", "2d"):
            os.environ["MVF_CONK"] = "rows" if form.startswith("rows") else form
            os.environ.pop("MVF_CONK_ROWS", None)
            os.environ.pop("MVF_CONK_WIDE", None)
            if form.startswith("rows8"):
                os.environ["MVF_CONK_ROWS"] = "8"
            if form.endswith("_wide"):
                os.environ["MVF_CONK_WIDE"] = "1"
            f = lambda: L.check(lib.mvf_con_k(xs.data_ptr(), nk, cs.data_ptr(), mk, 3, 2.7e-6, K.data_ptr(), code, st))
            f(); torch.cuda.synchronize()
            ev = []
            for _ in range(7):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); f(); b.record(); ev.append((a, b))
            torch.cuda.synchronize()
            ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
            gbps = K.element_size() * nk * mk / ms / 1e6
            out[f"{nk}x{mk}_{dt}_{form}"] = dict(ms=ms, GBps=gbps, checksum=float(K[::997].double().sum()))
            print(f"{nk} x {mk} {dt} {form:5s}: {ms:.3f} ms  {gbps:.0f} GB/s", flush=True)
        os.environ.pop("MVF_CONK", None)
        os.environ.pop("MVF_CONK_ROWS", None)
        os.environ.pop("MVF_CONK_WIDE", None)
        del K, xs, cs
        torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
