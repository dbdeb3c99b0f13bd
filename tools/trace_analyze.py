"""Reads a rocprofv3 --kernel-trace CSV of bench.py and says where the GPU's time goes in the steady state: per kernel the summed
duration, and how the wall time divides by what is running (k per-base kernels at once, only bin-level kernels, nothing).
usage: trace_analyze.py kernel_trace.csv [first_fraction last_fraction]"""
import csv, sys, re
from collections import defaultdict
path = sys.argv[1]
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.75
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        m = re.search(r"\b(k_\w+)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"].split("(")[0][:30]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
# the bench's kernels only (skip the generators), window = a slice of the run in the middle (timed steps)
rows = [r for r in rows if not r[2].startswith(("k_synth",))]
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
w0, w1 = t_lo + (t_hi - t_lo) * f0, t_lo + (t_hi - t_lo) * f1
PERBASE = ("k_fasta_classify", "k_gc_joint_hist", "k_gc_hist", "k_value_hist8", "k_cap_compact_bin8", "k_rescale_compact_bin8", "k_n_transitions", "k_escape_hist")
dur = defaultdict(float); cnt = defaultdict(int)
ev = []
for a, b, nm in rows:
    a2, b2 = max(a, w0), min(b, w1)
    if b2 <= a2: continue
    dur[nm] += b2 - a2; cnt[nm] += 1
    ev.append((a2, 1, nm)); ev.append((b2, -1, nm))
ev.sort(key=lambda e: (e[0], e[1]))
wall = w1 - w0
state = defaultdict(float)
nper = 0; nother = 0; last = w0
for t, d, nm in ev:
    state[(min(nper, 3), min(nother, 4))] += t - last
    last = t
    if nm in PERBASE: nper += d
    else: nother += d
state[(min(nper, 3), min(nother, 4))] += w1 - last
print(f"window {wall/1e6:.1f} ms; kernel-time summed {sum(dur.values())/1e6:.1f} ms")
for nm in sorted(dur, key=lambda k: -dur[k]):
    print(f"  {nm:32s} {dur[nm]/1e6:8.2f} ms  {cnt[nm]:6d} launches  avg {dur[nm]/cnt[nm]/1e3:7.1f} us  ({100*dur[nm]/wall:5.1f} % of wall)")
print("wall time by (per-base kernels running [0-3+], other kernels running [0-4+]):")
for k in sorted(state):
    if state[k] / wall > 0.003:
        print(f"  per-base {k[0]} other {k[1]}: {100*state[k]/wall:5.1f} %")
