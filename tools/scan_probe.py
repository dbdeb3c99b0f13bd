"""What does the scan kernel's time depend on?  One chromosome, with and without implanted events, short and long:
kernel times of one context (each launch alone on the chip)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsicnv_amd import api, synth
lib = api.load_library()
torch.cuda.set_device(0)
params = api.make_params(**synth.config_flags(4))
pool = api.RsiPool(0, 1)
pool.set_timing(1)
for n, nev, maxlen in ((125_000_000, 20, 100000), (125_000_000, 0, 100000), (125_000_000, 20, 10000), (125_000_000, 100, 100000), (30_000_000, 20, 100000), (30_000_000, 0, 100000)):
    p = synth.make_plan(n, 0x5EED0004 + 7, model=1, mean=30.0, n_events=nev, gaps=2, max_len=maxlen)
    d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr()); torch.cuda.synchronize()
    arg = [(d_rd.data_ptr(), d_fa.data_ptr(), p["n"])]
    for _ in range(2):
        pool.run(params, arg)
    pool.reset_times()
    reps = 5
    for _ in range(reps):
        r = pool.run(params, arg, collect_times=True)
    kt = pool.kernel_table()
    ev_bins = sum((b - a) for a, b, c in p["events"]) // 101
    print(f"n={n/1e6:.0f}Mb events={nev} (max {maxlen}, {ev_bins} bins in events) calls={len(r[0].calls('calls'))} Lmax={r[0].stats['Lmax']} tiles listed {r[0].stats['scan_tiles_listed']} of 2x{r[0].stats['scan_tiles']}: "
          + ", ".join(f"{k}={kt[k][0]/kt[k][1]*1e3:.0f}us x{kt[k][1]//reps}" for k in ("rsi_scan", "level_stop", "resolve_runs", "level_sums", "hist_walk", "minmax_plan", "candidate_test", "best_subsegment") if k in kt), flush=True)
    del d_fa, d_rd
# configs[4]: 60x, -m 51 -MED -cap 4
params5 = api.make_params(**synth.config_flags(5))
for chrom in (12, 20):
    p = synth.config_plan(5, chrom=chrom)
    d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr()); torch.cuda.synchronize()
    arg = [(d_rd.data_ptr(), d_fa.data_ptr(), p["n"])]
    for _ in range(2):
        pool.run(params5, arg)
    pool.reset_times()
    reps = 5
    for _ in range(reps):
        r = pool.run(params5, arg, collect_times=True)
    kt = pool.kernel_table()
    st = r[0].stats
    print(f"configs[4] chr{chrom+1} n={p['n']/1e6:.0f}Mb Lmax={st['Lmax']} tiles listed {st['scan_tiles_listed']} of 2x{st['scan_tiles']} tmedian1={st['tmedian1']} tlamda1={st['tlamda1']:.3f}: "
          + ", ".join(f"{k}={kt[k][0]/kt[k][1]*1e3:.0f}us x{kt[k][1]//reps}" for k in ("rsi_scan", "level_stop", "resolve_runs", "level_sums", "hist_walk", "minmax_plan", "candidate_test", "trim_runs") if k in kt), flush=True)
    del d_fa, d_rd
pool.close()
