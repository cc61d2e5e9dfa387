"""Developer probe (GPU): why is con_K slower into torch-allocated memory?  Address alignment experiment."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd import _lib as L
lib = L.load()
hip = ctypes.CDLL("libamdhip64.so")
nk, mk = 2_000_000, 2000
xs = torch.zeros(nk, 3, device="cuda"); cs = torch.zeros(mk, 3, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def run(p, name):
    f = lambda: L.check(lib.mvf_con_k(xs.data_ptr(), nk, cs.data_ptr(), mk, 3, 1e-5, p, 0, st))
    f(); torch.cuda.synchronize()
    ev = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in ev])
    print(f"{name:34s} ptr%2MiB={p % (2<<20):8d} ptr%4KiB={p % 4096:5d}  {ms:.3f} ms  {4.0*nk*mk/ms/1e6:.0f} GB/s")
K = torch.empty(nk * mk + (1 << 20), dtype=torch.float32, device="cuda")
run(K.data_ptr(), "torch.empty")
run(K.data_ptr() + 4096, "torch.empty + 4 KiB")
run(K.data_ptr() + (1 << 20), "torch.empty + 1 MiB")
del K; torch.cuda.empty_cache()
ptr = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(4 * nk * mk + (4 << 20))) == 0
run(ptr.value, "hipMalloc")
run(ptr.value + 512, "hipMalloc + 512 B")
run(ptr.value + 4096, "hipMalloc + 4 KiB")
K = torch.empty(nk * mk, dtype=torch.float32, device="cuda")
run(K.data_ptr(), "torch.empty (after hipMalloc)")
