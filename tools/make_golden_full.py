#!/usr/bin/env python3
"""Full-size golden call tables from the REAL reference (oracle/_ref/libref.so) for the BASELINE.json configurations the
GPU suite checks at their stated sizes (tests/test_full_size.py):

  cfg2_60mb           configs[1]: the 60 Mb Poisson chromosome, -m 101 -NB
  cfg3_250mb          configs[2]: the 250 Mb gamma-Poisson chromosome, -m 101 -NB
  cfg4_chr19/21/22    configs[3]: three chromosomes (<= 60 Mb) of the 24-chromosome 3 Gb genome, -m 101 -NB
  cfg4_chr8           configs[3]: a 142 Mb chromosome of the same genome (the in-flight check's large one)
  cfg5_chr13          configs[4]: one 60x chromosome of 112 Mb, -m 51 -MED -cap 4

Per case: the plan, sha256 of the generated inputs (guards against generator drift), chromosome median / SD, the padded N
regions, sha256 of the capped + compacted depth, and the raw / final call tables.  Runs only where /root/reference exists
(minutes of reference time in all); the files are data, never reference source.
  python tools/make_golden_full.py [NAME ...]"""
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


def main():
    import oracle
    from rsicnv_amd import api, synth
    lib = api.load_library()
    R = oracle.Ref()
    outdir = os.path.join(ROOT, "tests", "golden")
    for name in (sys.argv[1:] or list(CASES)):
        cfg, chrom = CASES[name]
        plan = synth.config_plan(cfg, chrom=chrom)
        flags = synth.config_flags(cfg)
        t0 = time.time()
        fasta, depth = synth.generate_host(lib, plan)
        p = oracle.make_params(**flags)
        R.load(p, depth, fasta)
        noncode = R.noncode()
        R.stage_gc()
        R.stage_cap()
        R.stage_concat()
        rd_concat = R.rd()
        rdmedian, rdsd = R.chrom_scalars()
        raw, fin, _ = R.detect()
        arr = lambda calls: np.array([[c[k] for k in CALL_KEYS] for c in calls], dtype=np.float64).reshape(len(calls), len(CALL_KEYS))
        np.savez_compressed(os.path.join(outdir, name + ".npz"), plan=json.dumps(plan), flags=json.dumps(flags), config=cfg, chrom=chrom,
                            fasta_sha=sha(fasta), depth_sha=sha(depth), noncode=np.asarray(noncode, dtype=np.int32),
                            rd_concat_sha=sha(rd_concat), n_compact=len(rd_concat), chrom_scalars=np.array([rdmedian, rdsd]),
                            calls_raw=arr(raw), calls=arr(fin))
        print(f"{name}: n={plan['n']} calls raw/final {len(raw)}/{len(fin)}, {time.time()-t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
