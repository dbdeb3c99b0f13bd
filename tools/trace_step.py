"""One genome step with RSI_HOT_TRACE=1: when every chromosome started and ended, and what its worker waited for (stderr of the library)."""
import os, sys, time
os.environ["RSI_HOT_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsicnv_amd import api, synth
lib = api.load_library(); torch.cuda.set_device(0)
params = api.make_params(**synth.config_flags(4))
args = []
keep = []
for c in range(24):
    p = synth.config_plan(4, chrom=c)
    d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr()); keep.append((d_fa, d_rd))
    args.append((d_rd.data_ptr(), d_fa.data_ptr(), p["n"]))
torch.cuda.synchronize()
pool = api.RsiPool(0, 12); pool.set_timing(0)
for i in range(4):
    sys.stderr.write(f"=== step {i}\n"); sys.stderr.flush()
    pool.run(params, args)
