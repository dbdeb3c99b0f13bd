"""Quick health probe of a GPU box: host compute (1 and 12 threads), launch/sync latency, small and
large device->host copies.  Used to tell pool artefacts from real regressions."""
import time, threading, sys
import numpy as np
import torch

def cpu_spin(n=3_000_000):
    a = np.random.default_rng(1).random(n).astype(np.float32)
    t0 = time.perf_counter(); np.sort(a); return (time.perf_counter() - t0) * 1e3

def threads(k=12):
    out = [0.0] * k
    def w(i): out[i] = cpu_spin()
    th = [threading.Thread(target=w, args=(i,)) for i in range(k)]
    t0 = time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    return (time.perf_counter() - t0) * 1e3, max(out)

x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
def lat(f, n=300):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
l_sync = lat(lambda: (x.add_(1), torch.cuda.synchronize()))
l_item = lat(lambda: x[0].item())
big = torch.zeros(10 << 20 >> 2, device="cuda", dtype=torch.int32)
host = torch.empty_like(big, device="cpu")
pin = torch.empty_like(big, device="cpu").pin_memory()
def bw(dst):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dst.copy_(big)
    torch.cuda.synchronize(); return 10 * 10.0 / 1024 / (time.perf_counter() - t0)
one = cpu_spin()
wall, worst = threads()
print(f"[probe] sort3M 1thr {one:.1f} ms | 12thr wall {wall:.1f} worst {worst:.1f} | add+sync {l_sync:.1f} us | item {l_item:.1f} us | D2H 10MB pageable {bw(host):.2f} GB/s pinned {bw(pin):.2f} GB/s", flush=True)
