"""How much would back-to-back genomes overlap?  One pool call over the genome's chromosomes k times over
(k = 1, 2, 4): the tail of one copy runs next to the head of the next."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsicnv_amd import api, synth

lib = api.load_library()
torch.cuda.set_device(0)
params = api.make_params(**synth.config_flags(4))
data = []
for c in range(24):
    p = synth.config_plan(4, chrom=c)
    d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda")
    d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
    data.append((d_rd, d_fa, p["n"]))
torch.cuda.synchronize()
args = [(a.data_ptr(), b.data_ptr(), n) for a, b, n in data]
pool = api.RsiPool(0, int(os.environ.get("WORKERS", "12")))
pool.set_timing(0)
for k in (1, 2, 4, 1, 2, 4):
    for _ in range(2):
        pool.run(params, args * k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        pool.run(params, args * k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print(f"k={k}: {dt:.1f} ms per call, {dt / k:.2f} ms per genome", flush=True)
