#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_TAG/passN/*counter_collection.csv): per kernel,
the mean of every counter per dispatch.  FETCH_SIZE / WRITE_SIZE are in KB; per MI355X_MICROARCH.md
section HBM, FETCH_SIZE reads half of the bytes of a wide coalesced stream on gfx950, so the
corrected read traffic is 2 x FETCH_SIZE (WRITE_SIZE is exact for 16-byte streaming stores)."""
import csv
import re
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"\b(k_\w+)", row["Kernel_Name"])
            k = m.group(1) if m else row["Kernel_Name"].split("(")[0][:40]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
print("kernel," + ",".join(names) + ",launches")
for k in sorted(acc):
    n = max(v[1] for v in acc[k].values())
    print(k + "," + ",".join(f"{acc[k][c][0] / acc[k][c][1]:.4g}" if c in acc[k] else "" for c in names) + f",{n}")

# --json OUT BASES: also write the per-launch FETCH/WRITE sizes of the streaming kernels for bench.py
if "--json" in sys.argv:
    import json
    out, bases = sys.argv[sys.argv.index("--json") + 1], int(sys.argv[sys.argv.index("--json") + 2])
    names_map = {"k_gc_hist": "gc_hist", "k_gc_rescale": "gc_rescale", "k_cap_compact_bin": "cap_compact_bin_int32",
                 "k_cap_compact_bin8": "cap_compact_bin_r2", "k_rescale_compact_bin8": "cap_compact_bin_k4j", "k_rescale_compact_stream": "cap_compact_bin",
                 "k_bin_median8": "bin_median", "k_value_hist8": "value_hist8",
                 "k_fasta_classify": "fasta_classify", "k_gc_joint_hist": "gc_joint_hist", "k_rsi_scan": "rsi_scan", "k_scan_detect": "scan_detect"}
    d = {"source": os.path.basename(root.rstrip("/")), "bases_per_launch": bases, "kernels": {}}
    for k, short in names_map.items():
        if k in acc and "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
            d["kernels"][short] = {"FETCH_SIZE_KB": acc[k]["FETCH_SIZE"][0] / acc[k]["FETCH_SIZE"][1],
                                   "WRITE_SIZE_KB": acc[k]["WRITE_SIZE"][0] / acc[k]["WRITE_SIZE"][1]}
            for extra in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVES"):   # wave-instructions per launch (bench.py: the `valu` fraction)
                if extra in acc[k]:
                    d["kernels"][short][extra] = acc[k][extra][0] / acc[k][extra][1]
    with open(out, "w") as f:
        json.dump(d, f, indent=1)
