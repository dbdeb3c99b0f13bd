"""Latency of one 250 Mb chromosome through pools of different sizes (who pays what)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsicnv_amd import api, synth
lib = api.load_library()
torch.cuda.set_device(0)
p = synth.config_plan(3, chrom=0)
d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr()); torch.cuda.synchronize()
params = api.make_params(**synth.config_flags(3))
for W, timing in ((12, False), (12, True), (1, False), (2, False), (4, False)):
    pool = api.RsiPool(0, W)
    pool.set_timing(timing)
    for i in range(5):
        pool.reset_times()
        t0 = time.perf_counter()
        r = pool.run(params, [(d_rd.data_ptr(), d_fa.data_ptr(), p["n"])], collect_times=True)
        dt = (time.perf_counter() - t0) * 1e3
        ph = sorted(pool.phase_table().items(), key=lambda kv: -kv[1])
        ph = [(k, round(v, 1)) for k, v in ph if not k.startswith(("calls.spec", "calls.nt", "calls.la"))][:6]
        print(f"W={W} timing={timing} run {dt:.1f} ms  {ph}", flush=True)
    pool.close()
