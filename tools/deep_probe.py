"""The same 60 Mb chromosome at 30x and scaled to 300x through one context: wall time, host phases and per-kernel times
(the deep-coverage path).  python tools/deep_probe.py"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsicnv_amd import api, synth
lib = api.load_library()
hot = api.RsiHot(0)
plan = synth.config_plan(2); n = plan["n"]
d_fa = torch.empty(n + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(n + 16, dtype=torch.int32, device="cuda")
synth.generate_device(lib, plan, d_fa.data_ptr(), d_rd.data_ptr())
g = torch.Generator(device="cuda").manual_seed(300)
d_deep = torch.where(d_rd > 0, d_rd * 10 + torch.randint(0, 10, d_rd.shape, device="cuda", dtype=torch.int32, generator=g), torch.zeros_like(d_rd))
p = api.make_params()
# bench.py's envelope case: the overdispersed model at a mean of 300 (2 % of the values outside K4w's window)
plan_nb = synth.make_plan(60_000_000, 0x5EED0E00, model=1, n_events=9, gaps=1, mean=300.0)
d_fa2 = torch.empty(plan_nb["n"] + 64, dtype=torch.uint8, device="cuda"); d_nb = torch.empty(plan_nb["n"] + 16, dtype=torch.int32, device="cuda")
synth.generate_device(lib, plan_nb, d_fa2.data_ptr(), d_nb.data_ptr())
for name, buf in (("30x", d_rd), ("300x", d_deep), ("300x gamma-Poisson", d_nb)):
    for timing in (0, 1):
        hot.set_timing(timing)
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fa = d_fa2 if buf is d_nb else d_fa
            r = hot.run_device(p, buf.data_ptr(), fa.data_ptr(), plan_nb["n"] if buf is d_nb else n)
            dt = time.perf_counter() - t0
        if timing == 0:
            print(name, f"{dt*1e3:.2f} ms", [(k, round(v, 2)) for k, v in hot.phase_times() if v > 0.05])
        else:
            print(name, "kernels", sorted([(k, round(v, 3)) for k, v in hot.kernel_times()], key=lambda kv: -kv[1])[:14])
