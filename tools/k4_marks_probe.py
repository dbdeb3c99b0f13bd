"""GC levels of a chromosome without a verified fixed-point ratio (the lanes within reach of one are left to K4s's exact pass).
usage: k4_marks_probe.py [CFG:CHR ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rsicnv_amd import api, synth
lib = api.load_library()
torch.cuda.set_device(0)
hot = api.RsiHot(0)
for spec in sys.argv[1:] or ["4:11", "3:0", "5:11", "4:0"]:
    cfg, chrom = (int(x) for x in spec.split(":"))
    p = synth.config_plan(cfg, chrom=chrom)
    d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
    torch.cuda.synchronize()
    res = hot.run_device(api.make_params(**synth.config_flags(cfg)), d_rd.data_ptr(), d_fa.data_ptr(), p["n"])
    rt = hot.fetch("k4_ratios").view(np.uint32)
    bad = [(g, hex(int(r))) for g, r in enumerate(rt) if (r >> 31) & 1 and not (r >> 30) & 1]
    print(f"cfg{cfg}/chr{chrom+1} n={p['n']}: levels occurring without a verified ratio: {bad[:12]}{' ...' if len(bad) > 12 else ''} ({len(bad)})", flush=True)
hot.close()
