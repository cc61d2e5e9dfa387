"""Where the wall time of one BASELINE config 5 organ fit goes (host profile), and the 4-organ run on 4 streams."""
import cProfile, pstats, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFC, SparseVFC_many
organs = [(*make_config("C2", N=250_000, seed=100 + s)[:2], None) for s in range(4)]
kw = dict(M=500, lambda_=0.02, MaxIter=30, dtype="float32")
SparseVFC_many(organs[:1], n_streams=1, **dict(kw, MaxIter=2))
for rep in range(2):
    t0 = time.perf_counter(); seq = [SparseVFC(*o, **kw) for o in organs]; t_seq = time.perf_counter() - t0
    t0 = time.perf_counter(); par = SparseVFC_many(organs, n_streams=4, **kw); t_par = time.perf_counter() - t0
    print(f"rep {rep}: sequential {t_seq*1e3:.1f} ms, 4 streams {t_par*1e3:.1f} ms, iterations {[int(v['iteration']) for v in par]}", flush=True)
pr = cProfile.Profile(); pr.enable()
SparseVFC(*organs[0], **kw); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38); print(s.getvalue()[:6000])
