#!/usr/bin/env python3
"""tests/golden/bam_small.npz: per-base depth the REAL reference (oracle/_ref/rsicnv_ref -b ... -s) produces for the
deterministic synthetic BAM of tests/bam_util.py, for two (minq, min_baseQ) settings and two chromosomes.
Data only: the plan of the synthetic reads + the reference's outputs."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bam_util as bu
import oracle
from conftest import make_case
from test_hot_extra import _write_case
from rsicnv_amd import api

lib = api.load_library()
spec = bu.golden_spec()
tmp = tempfile.mkdtemp(dir="/tmp")
out = {}
bam, refs, recs = bu.build_golden_bam(tmp)
libref = os.path.join(os.path.dirname(oracle.REF_BIN), "libref.so")
for chrom, n in refs:
    _, fasta, depth = make_case(lib, dict(n=n, seed=0xBA4 + len(chrom), model=0, n_events=2, gaps=1, max_len=8000, end_n=3000, gap_len=5000))
    d = os.path.join(tmp, chrom); os.makedirs(d, exist_ok=True)
    fa, _ = _write_case(d, fasta, depth, chrom=chrom)
    for q, Q in spec["settings"]:
        rd, _ = bu.reference_depth_dump(oracle.REF_BIN, libref, bam, fa, chrom, d, extra=("-q", str(q), "-Q", str(Q)))
        out[f"{chrom}_q{q}_Q{Q}"] = rd
        print(chrom, q, Q, "mean depth %.2f" % rd.mean())
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bam_small.npz"), **out)
print("written", os.path.getsize(os.path.join(ROOT, "tests", "golden", "bam_small.npz")) // 1024, "KB")
