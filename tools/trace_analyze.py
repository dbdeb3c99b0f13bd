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
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
rows.sort()
# the bench's kernels only (skip the generators), window = a slice of the run in the middle (timed steps)
rows = [r for r in rows if not r[2].startswith(("k_synth",))]
t_lo, t_hi = rows[0][0], max(r[1] for r in rows)
w0, w1 = t_lo + (t_hi - t_lo) * f0, t_lo + (t_hi - t_lo) * f1
PERBASE = ("k_fasta_classify", "k_gc_joint_hist", "k_gc_hist", "k_value_hist8", "k_cap_compact_bin8", "k_rescale_compact_bin8", "k_n_transitions", "k_escape_hist")
dur = defaultdict(float); cnt = defaultdict(int)
ev = []
for a, b, nm, _q, _s in rows:
    a2, b2 = max(a, w0), min(b, w1)
    if b2 <= a2: continue
    dur[nm] += b2 - a2; cnt[nm] += 1
    ev.append((a2, 1, nm)); ev.append((b2, -1, nm))
ev.sort(key=lambda e: (e[0], e[1]))
wall = w1 - w0
state = defaultdict(float)
conc = defaultdict(float)   # wall time by the number of kernels running at once
nper = 0; nother = 0; last = w0
for t, d, nm in ev:
    state[(min(nper, 3), min(nother, 4))] += t - last
    conc[nper + nother] += t - last
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
print("wall time by kernels running at once: " + "  ".join(f"{k}: {100*conc[k]/wall:.1f}%" for k in sorted(conc) if conc[k] / wall > 0.002)
      + f"   mean {sum(k * v for k, v in conc.items()) / wall:.2f}")
# per queue: consecutive kernels of one queue -- does a kernel's start precede its predecessor's end (the start stamp would then be
# the packet's, not the first wave's), and how long are the gaps between one kernel's end and the next one's start?
byq = defaultdict(list)
for a, b, nm, q, st in rows:
    if a >= w0 and b <= w1: byq[q].append((a, b, nm))
over = 0; pairs = 0; gaps = []
small = defaultdict(list)
for q, lst in byq.items():
    lst.sort()
    for (a0, b0, n0), (a1, b1, n1) in zip(lst, lst[1:]):
        pairs += 1
        if a1 < b0: over += 1
        gaps.append((a1 - b0) / 1e3)
        small[n1].append(((a1 - b0) / 1e3, (b1 - a1) / 1e3))
gaps.sort()
if gaps:
    q = lambda f: gaps[min(len(gaps) - 1, int(f * len(gaps)))]
    print(f"{len(byq)} queues; {pairs} consecutive pairs, {over} where the next kernel's start precedes the previous one's end; gap end -> next start (us): "
          f"p10 {q(0.1):.1f} p50 {q(0.5):.1f} p90 {q(0.9):.1f} mean {sum(gaps)/len(gaps):.1f}")
    for nm in ("k_trim_runs", "k_fs_chunk_sums", "k_run_prefix", "k_hist_walk", "k_minmax_plan", "k_n_transitions", "k_gc_joint_hist", "k_rescale_compact_bin8"):
        if small[nm]:
            g = sorted(x[0] for x in small[nm]); d = sorted(x[1] for x in small[nm])
            print(f"  {nm:26s} gap before: p50 {g[len(g)//2]:7.1f} us   duration: p50 {d[len(d)//2]:7.1f} us  (n={len(g)})")
