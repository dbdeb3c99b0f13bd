"""All host-view phases of one lone chromosome (configs[2], 250 Mb) through a one-worker pool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsicnv_amd import api, synth
lib = api.load_library()
torch.cuda.set_device(0)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = synth.config_plan(cfg, chrom=0)
d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr()); torch.cuda.synchronize()
params = api.make_params(**synth.config_flags(cfg))
pool = api.RsiPool(0, 1)
pool.set_timing(False)
for i in range(4):
    pool.reset_times()
    t0 = time.perf_counter()
    r = pool.run(params, [(d_rd.data_ptr(), d_fa.data_ptr(), p["n"])], collect_times=True)
    dt = (time.perf_counter() - t0) * 1e3
ph = [(k, round(v, 2)) for k, v in pool.phase_table().items() if not k.startswith(("calls.spec", "calls.nt", "calls.la"))]
print(f"run {dt:.2f} ms", ph, flush=True)
pool.set_timing(True)
pool.reset_times()
pool.run(params, [(d_rd.data_ptr(), d_fa.data_ptr(), p["n"])], collect_times=True)
print("kernels:", [(k, round(v[0], 3), int(v[1])) for k, v in sorted(pool.kernel_table().items(), key=lambda kv: -kv[1][0])], flush=True)
