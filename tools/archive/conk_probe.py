"""Developer probe (GPU): con_K bandwidth in a fresh process, zero vs realistic inputs."""
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
K = torch.empty(nk, mk, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(xs, cs, beta, name):
    f = lambda: L.check(lib.mvf_con_k(xs.data_ptr(), nk, cs.data_ptr(), mk, 3, beta, K.data_ptr(), 0, st))
    f(); torch.cuda.synchronize()
    ev = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in ev])
    print(f"{name:28s} {ms:.3f} ms  {4.0*nk*mk/ms/1e6:.0f} GB/s   mean K {float(K[:1000].mean()):.3g}")
z = torch.zeros(nk, 3, device="cuda"); zc = torch.zeros(mk, 3, device="cuda")
run(z, zc, 1e-5, "zeros")
xs = torch.from_numpy((X - ctrl.mean(0)).astype(np.float32)).cuda(); cs = torch.from_numpy((ctrl - ctrl.mean(0)).astype(np.float32)).cuda()
run(xs, cs, 2.7e-6, "real coords beta 2.7e-6")
run(xs, cs, 2.7e-5, "real coords beta 2.7e-5")
run(xs, cs, 2.7e-8, "real coords beta 2.7e-8")
run(z, zc, 1e-5, "zeros again")
# --- allocation source: raw hipMalloc vs torch caching allocator
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
ptr = ctypes.c_void_p()
del K; torch.cuda.empty_cache()
assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(4 * nk * mk)) == 0
def run_raw(name):
    f = lambda: L.check(lib.mvf_con_k(xs.data_ptr(), nk, cs.data_ptr(), mk, 3, 2.7e-6, ptr.value, 0, st))
    f(); torch.cuda.synchronize()
    ev = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in ev])
    print(f"{name:28s} {ms:.3f} ms  {4.0*nk*mk/ms/1e6:.0f} GB/s")
run_raw("raw hipMalloc output")
print("alloc conf:", os.environ.get("PYTORCH_HIP_ALLOC_CONF"), os.environ.get("PYTORCH_CUDA_ALLOC_CONF"), torch.cuda.memory.get_allocator_backend())
