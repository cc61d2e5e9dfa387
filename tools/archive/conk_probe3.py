import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
hip = ctypes.CDLL("libamdhip64.so")
libs = {"nontemporal": ctypes.CDLL(os.path.join(ROOT, "spateo-release_amd/spateo_amd/lib/libmvf.so")), "temporal": ctypes.CDLL(os.path.join(ROOT, "tools/libmvf_temporal.so"))}
for l in libs.values():
    l.mvf_con_k.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
nk, mk = 2_000_000, 2000
xs = torch.zeros(nk, 3, device="cuda"); cs = torch.zeros(mk, 3, device="cuda")
def run(p, name):
    for ln, lib in libs.items():
        f = lambda: lib.mvf_con_k(xs.data_ptr(), nk, cs.data_ptr(), mk, 3, 1e-5, p, 0, None)
        f(); torch.cuda.synchronize()
        ev = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); ev.append((a, b))
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        print(f"{name:30s} {ln:12s} {ms:.3f} ms  {4.0*nk*mk/ms/1e6:.0f} GB/s")
K = torch.empty(nk * mk, dtype=torch.float32, device="cuda"); run(K.data_ptr(), "torch.empty #1")
del K; torch.cuda.empty_cache()
ptr = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(4 * nk * mk)); run(ptr.value, "hipMalloc")
K = torch.empty(nk * mk, dtype=torch.float32, device="cuda"); run(K.data_ptr(), "torch.empty #2")
