"""Developer probe (GPU): BASELINE config 3's con_K run (2 M x 2000 float32, default store pattern) a few times - the
command the PMC passes wrap (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd import _lib as L
from spateo_amd._synthetic import make_config
lib = L.load()
nk, mk = 2_000_000, 2000
X, V, _ = make_config("C4", N=nk)
ctrl = X[np.random.default_rng(0).choice(nk, mk, replace=False)]
xs = torch.from_numpy((X - ctrl.mean(0)).astype(np.float32)).cuda()
cs = torch.from_numpy((ctrl - ctrl.mean(0)).astype(np.float32)).cuda()
K = torch.empty(nk, mk, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(4):
    L.check(lib.mvf_con_k(xs.data_ptr(), nk, cs.data_ptr(), mk, 3, 2.7e-6, K.data_ptr(), 0, st))
torch.cuda.synchronize()
print("algorithmic bytes per launch", 4.0 * (nk * mk + 3 * nk + 3 * mk))
