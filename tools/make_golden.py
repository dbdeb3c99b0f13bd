#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (oracle/_ref/libref.so, built by
oracle/Makefile.ref from /root/reference).  Runs only where /root/reference exists; the fixtures
it writes are data (seeded inputs' hashes + the reference's outputs), never reference source.

  python tools/make_golden.py            # all cases of tests/conftest.small_cases()
  python tools/make_golden.py wide       # tests/conftest.wide_scan_cases() (minutes: long scans)

Per case the file holds: the plan (JSON), sha256 of the generated FASTA / depth (guards against
generator drift), the padded N regions, sha256 of the per-base arrays after GC adjust / cap /
compaction (full arrays for the cases listed in FULL), the bin arrays, the three scan status
arrays, all scalars, and the segment / block / call lists.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FULL = {"poisson_tail7", "gampois_med_m51_cap4"}   # cases whose per-base arrays are stored in full
CALL_KEYS = ("start", "end", "type", "geno", "status", "length", "qscore", "score", "p1", "cnvmed", "cnvsd", "cnviqr",
             "refmed", "refsd", "refiqr")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def calls_array(calls):
    return np.array([[c[k] for k in CALL_KEYS] for c in calls], dtype=np.float64).reshape(len(calls), len(CALL_KEYS))


def main():
    import oracle
    from conftest import make_case, small_cases, wide_scan_cases
    from rsicnv_amd import api
    lib = api.load_library()
    R = oracle.Ref()
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    cases = wide_scan_cases() if (len(sys.argv) > 1 and sys.argv[1] == "wide") else small_cases()
    only = set(sys.argv[2:]) if len(sys.argv) > 2 else None   # python tools/make_golden.py small NAME [NAME ...]
    for name, plan_kw, flag_kw in cases:
        if only is not None and name not in only:
            continue
        plan, fasta, depth = make_case(lib, plan_kw)
        p = oracle.make_params(**flag_kw)
        R.load(p, depth, fasta)
        noncode = R.noncode()
        R.stage_gc()
        rd_gc = R.rd()
        R.stage_cap()
        rd_cap = R.rd()
        R.stage_concat()
        rd_concat = R.rd()
        rdmedian, rdsd = R.chrom_scalars()
        binmed, binmedint, binnb = R.stage_bins()
        out = dict(plan=json.dumps(plan), flags=json.dumps(flag_kw), fasta_sha=sha(fasta), depth_sha=sha(depth),
                   noncode=noncode, rd_gc_sha=sha(rd_gc), rd_cap_sha=sha(rd_cap), rd_concat_sha=sha(rd_concat),
                   chrom=np.array([rdmedian, rdsd]), binmed=binmed, binmedint=binmedint, binnb=binnb)
        if name in FULL:
            out.update(rd_gc=rd_gc.astype(np.int32), rd_concat=rd_concat.astype(np.int32))
        trans = flag_kw.get("trans", 0)
        for use_med in ([True] if trans == 1 else [False] if trans == 0 else [True, False]):
            sc, st, segs = R.scan(use_med)
            assert sc.stepwise_matches_reference == 1, "stage driver diverged from the reference's own scan driver"
            pre = "med" if use_med else "nb"
            out[f"{pre}_scan"] = np.array([sc.tmedian1, sc.tsigma1, sc.tlamda1, sc.tmedian2, sc.tsigma2, sc.tlamda2,
                                           sc.target_tlamda, sc.Lmax, sc.cal_max])
            out[f"{pre}_status1"], out[f"{pre}_status1f"], out[f"{pre}_status2"] = st
            out[f"{pre}_segs"] = calls_array(segs)
        raw, fin, txt = R.detect()
        out["calls_raw"] = calls_array(raw)
        out["calls"] = calls_array(fin)
        out["rows"] = txt
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
        print(f"{name}: n={plan['n']} calls raw/final {len(raw)}/{len(fin)} -> {os.path.getsize(os.path.join(outdir, name + '.npz'))/1024:.0f} KB")


if __name__ == "__main__":
    main()
