"""Depth text ingestion rate: write a "pos depth" file for an N Mb chromosome, then time
 (a) the device loader (rsi_hot_load_depth_text: read -> pinned -> HBM -> parse kernel),
 (b) the sequential host loop it replaces in this repo (forced by shuffling two lines: same code path as the fallback).
Usage: python tools/text_ingest_bench.py [Mb=60]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rsicnv_amd import api

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 60
n = int(mb * 1e6)
rng = np.random.default_rng(1)
depth = rng.poisson(30, n).astype(np.int32)
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "depth.txt")
t0 = time.time()
pos = np.arange(1, n + 1)
# fast writer: fixed-width formatting through numpy char arrays would be quicker still; this is test plumbing
with open(path, "w") as f:
    step = 2_000_000
    for a in range(0, n, step):
        b = min(n, a + step)
        f.write("\n".join(f"{p}\t{v}" for p, v in zip(pos[a:b].tolist(), depth[a:b].tolist())) + "\n")
size = os.path.getsize(path)
print(f"wrote {size/1e6:.0f} MB of text for {n/1e6:.0f} Mb in {time.time()-t0:.0f} s", flush=True)
h = api.RsiHot(0)
h.set_timing(True)
for rep in range(3):
    st = h.load_depth_text(path, n)
    print(f"device loader: {st['t_total_ms']:.0f} ms total = {size/st['t_total_ms']/1e6:.2f} GB/s of text, {n/st['t_total_ms']/1e3:.1f} Mbases/s; "
          f"parse kernels {st['t_parse_kernel_ms']:.1f} ms = {size/max(st['t_parse_kernel_ms'],1e-9)/1e6:.0f} GB/s; lines {st['lines']} fallback {st['fallback']}", flush=True)
got = h.fetch("depth_in")
exp = depth.copy(); exp[-1] = 0
assert np.array_equal(got, exp), "device-parsed depth differs"
# unsorted variant -> the host loop
with open(path, "a") as f:
    f.write("5\t1\n")
st = h.load_depth_text(path, n)
print(f"host loop (fallback={st['fallback']}): {st['t_total_ms']:.0f} ms total = {size/st['t_total_ms']/1e6:.2f} GB/s of text, {n/st['t_total_ms']/1e3:.1f} Mbases/s", flush=True)
os.remove(path); os.rmdir(d)
