"""Latency of one chromosome through a stand-alone context (no pool): configs[1] and configs[2]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsicnv_amd import api, synth
lib = api.load_library(); torch.cuda.set_device(0)
h = api.RsiHot(0)
for cfg in (2, 3):
    p = synth.config_plan(cfg)
    d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr()); torch.cuda.synchronize()
    params = api.make_params(**synth.config_flags(cfg))
    for _ in range(3):
        h.run_device(params, d_rd.data_ptr(), d_fa.data_ptr(), p["n"])
    t0 = time.perf_counter()
    for _ in range(8):
        r = h.run_device(params, d_rd.data_ptr(), d_fa.data_ptr(), p["n"])
    dt = (time.perf_counter() - t0) / 8 * 1e3
    print(f"config {cfg}: {dt:.2f} ms, {len(r.calls('calls'))} calls, split={os.environ.get('RSI_HOT_CAND_SPLIT', 'auto')}", flush=True)
