#!/usr/bin/env python3
"""Full-size golden call tables from the REAL reference (oracle/_ref/libref.so) for the BASELINE.json configurations the
GPU suite checks at their stated sizes (tests/test_full_size.py):

  cfg2_60mb           configs[1]: the 60 Mb Poisson chromosome, -m 101 -NB
  cfg3_250mb          configs[2]: the 250 Mb gamma-Poisson chromosome, -m 101 -NB
  cfg4_chr19/21/22    configs[3]: three chromosomes (<= 60 Mb) of the 24-chromosome 3 Gb genome, -m 101 -NB
  cfg4_chr8           configs[3]: a 142 Mb chromosome of the same genome (the in-flight check's large one)
  cfg5_chr13          configs[4]: one 60x chromosome of 112 Mb, -m 51 -MED -cap 4
  genome4 / genome5   ALL 24 chromosomes of configs[3] / configs[4] (cfg4_chr1..24, cfg5_chr1..24), one reference
                      process per chromosome, `--jobs` at a time (the reference keeps its state in globals); then
                      tests/golden/genome_rows.json = per config the sha256 bench.py computes over a step's rows +
                      (chromosome, RDmedian, RDsd), from the REFERENCE's rows, so the timed work is pinned too

Per case: the plan, sha256 of the generated inputs (guards against generator drift), chromosome median / SD, the padded N
regions, sha256 of the capped + compacted depth, and the raw / final call tables.  Runs only where /root/reference exists
(minutes of reference time in all); the files are data, never reference source.
  python tools/make_golden_full.py [NAME ...]
  python tools/make_golden_full.py genome4 genome5 --jobs 5
  python tools/make_golden_full.py --rows-json          (only rebuild genome_rows.json from the .npz files)"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CALL_KEYS = ("start", "end", "type", "geno", "status", "length", "qscore", "score", "p1", "cnvmed", "cnvsd", "cnviqr",
             "refmed", "refsd", "refiqr")
CASES = {"cfg2_60mb": (2, 0), "cfg3_250mb": (3, 0), "cfg4_chr19": (4, 18), "cfg4_chr21": (4, 20), "cfg4_chr22": (4, 21), "cfg5_chr13": (5, 12),
         "cfg4_chr8": (4, 7)}   # a chromosome of 140 Mb and more for the in-flight check


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


for _c in range(24):
    CASES[f"cfg4_chr{_c + 1}"] = (4, _c)
    CASES[f"cfg5_chr{_c + 1}"] = (5, _c)
OUTDIR = os.path.join(ROOT, "tests", "golden")


def genome_rows_hash(cfg):
    """bench.py's step hash (finish_step) computed from the reference's rows: every row + newline in chromosome order, then
    repr((chromosome index, RDmedian, RDsd)) per chromosome."""
    h = hashlib.sha256()
    stats, ncalls = [], 0
    for c in range(24):
        g = np.load(os.path.join(OUTDIR, f"cfg{cfg}_chr{c + 1}.npz"), allow_pickle=False)
        for r in str(g["rows"]).splitlines():
            h.update(r.encode()); h.update(b"\n")
            ncalls += 1
        stats.append((c, float(g["chrom_scalars"][0]), float(g["chrom_scalars"][1])))
    for s in stats:
        h.update(repr(s).encode())
    return {"rows_sha256": h.hexdigest(), "calls": ncalls, "chromosomes": [list(s) for s in stats]}


def write_rows_json():
    out = {}
    for cfg in (4, 5):
        if all(os.path.exists(os.path.join(OUTDIR, f"cfg{cfg}_chr{c + 1}.npz")) for c in range(24)):
            out[f"config{cfg}"] = genome_rows_hash(cfg)
    with open(os.path.join(OUTDIR, "genome_rows.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("genome_rows.json:", {k: (v["rows_sha256"][:16], v["calls"]) for k, v in out.items()})


def main():
    argv = sys.argv[1:]
    jobs = 4
    if "--jobs" in argv:
        i = argv.index("--jobs")
        jobs = int(argv[i + 1])
        del argv[i:i + 2]
    if "--rows-json" in argv:
        return write_rows_json()
    genomes = [a for a in argv if a in ("genome4", "genome5")]
    if genomes:   # one process per chromosome, longest first
        import subprocess
        from rsicnv_amd import synth
        names = [f"cfg{g[-1]}_chr{c + 1}" for g in genomes for c in range(24)]
        names.sort(key=lambda nm: -synth.config_plan(*CASES[nm][:1], chrom=CASES[nm][1])["n"])
        running = []
        while names or running:
            while names and len(running) < jobs:
                nm = names.pop(0)
                running.append((nm, subprocess.Popen([sys.executable, os.path.abspath(__file__), nm])))
            time.sleep(1.0)
            for nm, pr in list(running):
                if pr.poll() is not None:
                    if pr.returncode != 0:
                        raise SystemExit(f"{nm}: exit {pr.returncode}")
                    running.remove((nm, pr))
        return write_rows_json()
    import oracle
    from rsicnv_amd import api, synth
    lib = api.load_library()
    R = oracle.Ref()
    outdir = OUTDIR
    for name in (argv or ["cfg2_60mb", "cfg3_250mb"]):
        cfg, chrom = CASES[name]
        plan = synth.config_plan(cfg, chrom=chrom)
        flags = synth.config_flags(cfg)
        t0 = time.time()
        fasta, depth = synth.generate_host(lib, plan)
        p = oracle.make_params(**flags)
        R.load(p, depth, fasta, chrom=f"chr{chrom + 1}" if cfg in (4, 5) else "chrS")
        noncode = R.noncode()
        R.stage_gc()
        R.stage_cap()
        R.stage_concat()
        rd_concat = R.rd()
        rdmedian, rdsd = R.chrom_scalars()
        raw, fin, rows = R.detect()
        arr = lambda calls: np.array([[c[k] for k in CALL_KEYS] for c in calls], dtype=np.float64).reshape(len(calls), len(CALL_KEYS))
        np.savez_compressed(os.path.join(outdir, name + ".npz"), plan=json.dumps(plan), flags=json.dumps(flags), config=cfg, chrom=chrom,
                            fasta_sha=sha(fasta), depth_sha=sha(depth), noncode=np.asarray(noncode, dtype=np.int32),
                            rd_concat_sha=sha(rd_concat), n_compact=len(rd_concat), chrom_scalars=np.array([rdmedian, rdsd]),
                            calls_raw=arr(raw), calls=arr(fin), rows=rows)
        print(f"{name}: n={plan['n']} calls raw/final {len(raw)}/{len(fin)}, {time.time()-t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
