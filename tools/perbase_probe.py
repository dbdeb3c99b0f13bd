"""Per-base kernels alone on the chip under debug switches (RSI_HOT_K2J_DBG, RSI_HOT_K4J_DBG: ablations that break the
results -- a failing run is expected there; the kernel times are still read).  usage: perbase_probe.py "ENV=VAL,..." ..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsicnv_amd import api, synth
lib = api.load_library()
torch.cuda.set_device(0)
cases = []
sel = os.environ.get("PERBASE_CASES")   # e.g. "4:11" or "3:0,5:11": (config, chromosome index) pairs
case_list = [tuple(int(x) for x in c.split(":")) for c in sel.split(",")] if sel else [(3, 0), (4, 11), (5, 11)]
for cfg, chrom in case_list:
    p = synth.config_plan(cfg, chrom=chrom)
    d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
    cases.append((f"cfg{cfg}/chr{chrom+1} {p['n']/1e6:.0f}Mb", api.make_params(**synth.config_flags(cfg)), d_rd, d_fa, p["n"]))
torch.cuda.synchronize()
hot = api.RsiHot(0)
hot.set_timing(2)
for variant in sys.argv[1:] or [""]:
    for kv in variant.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            os.environ[k] = v
    for name, params, d_rd, d_fa, n in cases:
        acc = {}
        reps = 6
        for it in range(reps + 2):
            try:
                hot.run_device(params, d_rd.data_ptr(), d_fa.data_ptr(), n)
                ok = True
            except api.RsiError as e:
                ok = False
            if it < 2:
                continue
            for k, ms in hot.kernel_times():
                acc.setdefault(k, []).append(ms)
        print(f"[{variant or 'default'}] {name} ok={ok}: " + ", ".join(f"{k}={sum(v)/len(v)*1e3:.0f}us" for k, v in acc.items()), flush=True)
    for kv in variant.split(","):
        if "=" in kv:
            os.environ.pop(kv.split("=", 1)[0], None)
hot.close()
