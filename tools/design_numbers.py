#!/usr/bin/env python3
"""The numbers table of DESIGN.md section 7, generated from the committed profiles (one table built from
profiles/r5_* instead of prose copies).

  python tools/design_numbers.py            print the table
  python tools/design_numbers.py --write    replace the block between the markers in DESIGN.md
  python tools/design_numbers.py --check    exit 1 when DESIGN.md's block differs from what the profiles give (tests/test_abi.py)

Every cell names nothing but what a file under profiles/ holds; the column of the previous round comes from that round's files.
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
BEGIN, END = "<!-- numbers:begin (tools/design_numbers.py --write) -->", "<!-- numbers:end -->"


def line(name):
    p = os.path.join(PROF, name)
    if not os.path.exists(p):
        return None
    rows = [l for l in open(p).read().splitlines() if l.startswith("{")]
    return json.loads(rows[-1]) if rows else None


def kernel_avgs(name):
    """kernel -> (average us, calls) of a rocprofv3 kernel_stats csv."""
    p = os.path.join(PROF, name)
    out = {}
    if not os.path.exists(p):
        return out
    for r in csv.DictReader(open(p)):
        m = re.search(r"\b(k_\w+)", r["Name"])
        k = m.group(1) if m else r["Name"].split("(")[0]
        n, tot = int(r["Calls"]), float(r["TotalDurationNs"])
        if k in out:
            tot += out[k][0] * out[k][1] * 1e3
            n += out[k][1]
        out[k] = (tot / n / 1e3, n)
    return out


def pmc(name):
    p = os.path.join(PROF, name)
    if not os.path.exists(p):
        return {}
    return {r["kernel"]: r for r in csv.DictReader(open(p))}


def f(x, nd=2):
    return "–" if x is None else f"{x:.{nd}f}"


def get(d, *path):
    for k in path:
        if d is None:
            return None
        if isinstance(k, str) and isinstance(d, dict):
            d = d.get(k)
        elif isinstance(k, int) and isinstance(d, (list, tuple)):
            d = d[k] if k < len(d) else None
        else:
            return None
    return d


def first_key(d, prefix):
    if not d:
        return None
    for k in d:
        if k.startswith(prefix):
            return d[k]
    return None


def bench_cells(b):
    """the cells one bench line contributes, by row key."""
    c = {}
    if b is None:
        return c
    c["step"] = f"{b['ms_per_step']:.2f} ms, {b['value'] / 1e9:.1f} Gbases/s ({get(b, 'config', 'workers') or 16} workers, {get(b, 'config', 'steps_in_flight')} genomes queued); rows = the reference's rows: {b.get('rows_match_reference')}"
    c["one"] = f"{f(get(b, 'one_genome_at_a_time', 'ms_per_step'))} ms"
    r = b["roofline"]
    c["whole"] = f"{f(get(r, 'whole_path', 'frac'), 3)}"
    c["k2j"] = (f"{f(r['frac'], 3)} / {f(get(r, 'isolated', 'frac'), 3)} ({f(r['avg_launch_ms'] * 1e3, 0)} / {f(get(r, 'isolated', 'avg_launch_ms') * 1e3, 0)} µs per 125 Mb); "
                f"per launch {f(r['traffic'] / 1e6, 0) if r.get('traffic') else '–'} MB of PMC traffic for {f(r['algorithmic_bytes_per_launch'] / 1e6, 0)} MB algorithmic")
    sk = r.get("streaming_kernels", {})
    k4 = get(sk, "cap_compact_bin", "isolated", "avg_launch_ms")
    km = get(sk, "bin_median", "isolated", "avg_launch_ms")
    if k4 is not None:
        c["k4"] = f"{f(k4 * 1e3, 0)} µs" + (f" + {f(km * 1e3, 0)} µs (stream + medians) = {f((k4 + km) * 1e3, 0)} µs" if km is not None else " (one kernel)") + " per 125 Mb, alone"
    s = b.get("single_chromosome_configs") or {}
    c["lone"] = " / ".join(f(get(first_key(s, p), "ms"), 2) for p in ("configs[1]", "configs[2]")) + " ms"
    h = b.get("t_device_h2d") or {}
    c["h2d"] = " / ".join(f(get(first_key(h, p), "ms"), 1) for p in ("configs[1]", "configs[2]")) + " ms"
    g = b.get("t_device_h2d_genome") or {}
    if g:
        c["h2dg"] = f"{f(g.get('ms'), 0)} ms, {f(g.get('bases_per_s', 0) / 1e9, 1)} Gbases/s" + (f" ({g['link_bytes_per_base']:.0f} B/base over the link)" if g.get("link_bytes_per_base") else " (5 B/base over the link)")
    c["e2e"] = f"{f(get(b, 't_e2e', 's'), 2)} s" + (f"; configs[2] (250 Mb, 3.1 GB of text): {f(get(b, 't_e2e_250Mb', 's'), 2)} s = {f(get(b, 't_e2e_250Mb', 'bases_per_s') / 1e6, 0)} Mbases/s" if b.get("t_e2e_250Mb") else "")
    env = b.get("fallback_envelope") or {}
    base = get(first_key(env, "30x default"), "ms")
    if base:
        parts = [f"byte path {f(base)} ms"]
        for key, label in (("-NOGC", "`-NOGC`"), ("-m 201", "`-m 201`"), ("300x", "300×")):
            v = get(first_key(env, key), "ms")
            if v:
                parts.append(f"{label} {f(v)} ({v / base:.2f}×)")
        c["env"] = "; ".join(parts)
    side = b.get("configs[4]_side_pass") or {}
    if side:
        c["cfg5side"] = f"{f(side.get('ms_per_step'))} ms (side pass of the default line; rows = the reference's rows: {side.get('rows_match_reference')})"
    cb = b.get("cpu_baseline") or {}
    if cb:
        c["cpu"] = f"{cb['value'] / 1e6:.1f} Mbases/s on one core, {get(cb, 'all_cores', 'value') / 1e6:.0f} on {get(cb, 'all_cores', 'cores')}"
    return c


def trace_cells(ks, per_genome=24):
    c = {}
    if not ks:
        return c
    def us(k):
        return ks[k][0] if k in ks else None
    k4 = [us(k) for k in ("k_rescale_compact_stream", "k_bin_median8") if us(k) is not None] or [us("k_rescale_compact_bin8")]
    c["w1"] = (f"K2j {f(us('k_gc_joint_hist'), 0)}, K4 {' + '.join(f(x, 0) for x in k4)}, K1 {f(us('k_fasta_classify'), 0)}, K1b {f(us('k_n_transitions'), 0)} µs per launch "
               f"(= {f(sum(x for x in [us('k_gc_joint_hist'), us('k_fasta_classify'), us('k_n_transitions')] + k4 if x) * per_genome / 1e3)} ms per genome)")
    cand = [us(k) for k in ("k_cand_gather", "k_cand_prefix", "k_cand_means", "k_cand_hist")]
    if all(x is not None for x in cand):
        c["cand"] = f"{f(sum(cand) * per_genome / 1e3)} ms per genome (gather {f(cand[0], 0)}, prefix {f(cand[1], 0)}, means {f(cand[2], 0)}, hist {f(cand[3], 0)} µs per launch)"
    return c


def pmc_cells(p, bases=250e6):
    c = {}
    def per_base(k, col, mul=1.0):
        return float(p[k][col]) * mul / bases if k in p and p[k].get(col) else None
    rows = []
    for k, label in (("k_gc_joint_hist", "K2j"), ("k_rescale_compact_stream", "K4s"), ("k_bin_median8", "K4m"), ("k_rescale_compact_bin8", "K4j"), ("k_fasta_classify", "K1")):
        if k not in p:
            continue
        rd, wr, va = per_base(k, "FETCH_SIZE", 2048.0), per_base(k, "WRITE_SIZE", 1024.0), per_base(k, "SQ_INSTS_VALU", 64.0)
        rows.append(f"{label} {f(rd + wr)} B/base ({f(rd)} read + {f(wr)} written), {f(va, 1)} vector instructions per base")
    if rows:
        c["pmc"] = "; ".join(rows)
    return c


def rank_probe(name):
    p = os.path.join(PROF, name)
    if not os.path.exists(p):
        return None
    t = open(p).read()
    m = {int(a): b for a, b in re.findall(r"world (\d+):.*?-> x([\d.]+)", t)}
    one = re.search(r"world 1: ([\d.]+) ms", t)
    return "N = 2 / 4 / 8: " + " / ".join(f"{m[n]}×" for n in (2, 4, 8) if n in m) + (f" (whole genome {one.group(1)} ms on that box)" if one else "") if m else None


def table():
    prev, cur = bench_cells(line("r4_bench.json")), bench_cells(line("r5_bench.json"))
    for tag, cells in (("r4", prev), ("r5", cur)):
        cells.update(trace_cells(kernel_avgs(f"{tag}_kernel_stats_3Gb_w1.csv")))
        cells.update(pmc_cells(pmc(f"{tag}_pmc_summary_250Mb.csv")))
        b5 = line(f"{tag}_bench_config5.json")
        if b5:
            cells["cfg5"] = f"{b5['ms_per_step']:.2f} ms, {b5['value'] / 1e9:.0f} Gbases/s, whole path {f(get(b5, 'roofline', 'whole_path', 'frac'), 2)}; rows = the reference's rows: {b5.get('rows_match_reference')}"
        rp = rank_probe(f"{tag}_rank_probe.txt")
        if rp:
            cells["rank"] = rp
    reh = None
    p = os.path.join(PROF, "r5_sharded_rehearsal.json")
    if os.path.exists(p):
        runs = json.load(open(p))["runs"]
        reh = "; ".join(f"{k}: rows = the reference's rows: {v['rows_match_reference']}, {v['workers']} workers per rank, {v['ms_per_step']:.0f} ms per genome with all ranks on ONE GPU" for k, v in runs.items())
    e2e = None
    p = os.path.join(PROF, "r5_e2e_genome.json")
    if os.path.exists(p):
        e = json.load(open(p))
        e2e = (f"{e['chromosomes_run']} of {e['chromosomes_of_genome']} chromosomes, {e['depth_text_bytes'] / 1e9:.1f} GB of depth text: {e['sequential']['s']:.1f} s one after the other = "
               f"{e['sequential']['bases_per_s'] / 1e6:.0f} Mbases/s; four processes at once {e['4_processes_at_once']['s']:.1f} s = {e['4_processes_at_once']['bases_per_s'] / 1e6:.0f} Mbases/s (north_star: ≥ 50)")
    rows = [
        ("step", "3 Gb genome (configs[3]) through the pool, `python bench.py`"),
        ("one", "the same, one genome at a time (`one_genome_at_a_time`)"),
        ("whole", "whole path, fraction of 23.7 B/base × 8 TB/s"),
        ("k2j", "dominant per-base kernel K2j: `roofline.frac` in situ / `isolated`"),
        ("k4", "K4 (`streaming_kernels.cap_compact_bin` + `bin_median`, isolated)"),
        ("w1", "per-base kernels alone on the chip (the genome through ONE worker, `*_kernel_stats_3Gb_w1.csv`)"),
        ("pmc", "PMC, 250 Mb chromosome (`*_pmc_summary_250Mb.csv`; FETCH_SIZE doubled as the guide prescribes)"),
        ("cand", "candidate kernels in the one-worker trace"),
        ("cfg5", "configs[4] (`--config 5`: 60×, `-m 51 -MED -cap 4`), `*_bench_config5.json`"),
        ("cfg5side", "configs[4] as the default line's side pass"),
        ("lone", "configs[1] / configs[2] alone through the pool (`single_chromosome_configs`)"),
        ("rank", "a rank's share of the sharded genome on one GPU (`tools/rank_probe.py`)"),
        ("h2d", "`t_device_h2d` configs[1] / configs[2] (pinned host arrays → results on the host)"),
        ("h2dg", "the genome from pinned host memory (`t_device_h2d_genome`)"),
        ("e2e", "`t_e2e` configs[1]: depth text + FASTA → output file, one process"),
        ("env", "fallback envelope, one 60 Mb chromosome (`fallback_envelope`)"),
        ("cpu", "the compiled reference on the box's CPU (`cpu_baseline`)"),
    ]
    out = ["| | round 4 (`profiles/r4_*`) | round 5 (`profiles/r5_*`) |", "|---|---|---|"]
    for key, label in rows:
        a, b = prev.get(key), cur.get(key)
        if a is None and b is None:
            continue
        out.append(f"| {label} | {a or '–'} | {b or '–'} |")
    if e2e:
        out.append(f"| the 3 Gb genome end to end through `rsicnv rsi -d` (`tools/e2e_genome.py`, `profiles/r5_e2e_genome.json`) | – | {e2e} |")
    if reh:
        out.append(f"| sharded rehearsal at full size over gloo (`profiles/r5_sharded_rehearsal.json`) | – | {reh} |")
    return "\n".join(out)


def main():
    t = table()
    path = os.path.join(ROOT, "DESIGN.md")
    if "--write" in sys.argv or "--check" in sys.argv:
        s = open(path).read()
        if BEGIN not in s or END not in s:
            sys.exit("DESIGN.md has no numbers block")
        a, b = s.index(BEGIN) + len(BEGIN), s.index(END)
        if "--check" in sys.argv:
            if s[a:b].strip() != t.strip():
                sys.exit("DESIGN.md's numbers block is stale: run python tools/design_numbers.py --write")
            return
        open(path, "w").write(s[:a] + "\n" + t + "\n" + s[b:])
        return
    print(t)


if __name__ == "__main__":
    main()
