"""Host profile of a whole SparseVFC call (host arrays in, host dict out) at 8 M cells x 3000 control points, float32 cells."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
import spateo_amd as st
from spateo_amd._synthetic import make_config
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
X, V, M = make_config("C4", N=n)
kw = dict(M=M, lambda_=0.02, lstsq_method="scipy", seed=0, dtype="float32", device="cuda:0")
st.SparseVFC(X[:300000], V[:300000], None, **dict(kw, MaxIter=2))
for mode in ("full",):
    t0 = time.perf_counter(); r = st.SparseVFC(X, V, None, gram_mode=mode, **kw); torch.cuda.synchronize()
    print(f"{mode}: whole call {time.perf_counter() - t0:.2f} s, {int(r['iteration']) + 1} iterations")
pr = cProfile.Profile(); pr.enable(); r = st.SparseVFC(X, V, None, **kw); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:4200])
