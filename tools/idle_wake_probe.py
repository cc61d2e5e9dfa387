"""Does the device answer late after the host has left it idle for a few milliseconds?  (host sleeps X ms, then: a trivial kernel +
synchronise; a 6 MB pinned upload + synchronise; 30 repetitions each: median / 90th percentile / max in ms)"""
import time, numpy as np, torch
d = torch.device("cuda:0")
a = torch.zeros(1024, device=d)
pin = torch.empty(250_000 * 3, dtype=torch.float64, pin_memory=True)
dev = torch.empty_like(pin, device=d)
def stat(ts):
    ts = np.sort(np.asarray(ts) * 1e3); return f"median {ts[len(ts) // 2]:.3f}, p90 {ts[int(0.9 * len(ts))]:.3f}, max {ts[-1]:.3f}"
for idle_ms in (0, 1, 3, 5, 10, 20, 50):
    t_k, t_c = [], []
    for rep in range(30):
        torch.cuda.synchronize(); time.sleep(idle_ms * 1e-3)
        t0 = time.perf_counter(); a.add_(1.0); torch.cuda.synchronize(); t_k.append(time.perf_counter() - t0)
        torch.cuda.synchronize(); time.sleep(idle_ms * 1e-3)
        t0 = time.perf_counter(); dev.copy_(pin, non_blocking=True); torch.cuda.synchronize(); t_c.append(time.perf_counter() - t0)
    print(f"idle {idle_ms:3d} ms: kernel + sync {stat(t_k)};  6 MB pinned H2D + sync {stat(t_c)}")
# busy host instead of sleeping (numpy work of ~5 ms), like the preprocessing between two fits
x = np.random.rand(250_000, 3)
t_k = []
for rep in range(30):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    while time.perf_counter() - t1 < 5e-3: np.sqrt((x * x).sum(1))
    t0 = time.perf_counter(); a.add_(1.0); torch.cuda.synchronize(); t_k.append(time.perf_counter() - t0)
print(f"busy host 5 ms: kernel + sync {stat(t_k)}")
