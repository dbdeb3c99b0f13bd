"""BAM -> depth rate: a synthetic paired-end BAM for an N Mb chromosome at COV x, then the library's
loader (host inflate threads + device pileup) timed.  Usage: python tools/bam_ingest_bench.py [Mb=10] [cov=30]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bam_util as bu
from rsicnv_amd import api

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 10
cov = float(sys.argv[2]) if len(sys.argv) > 2 else 30
n = int(mb * 1e6)
depth = np.full(n, cov, dtype=np.float64)
t0 = time.time()
recs = bu.paired_reads_following_depth(depth, n)
d = tempfile.mkdtemp(dir="/tmp")
bam = os.path.join(d, "x.bam")
bu.write_bam(bam, [("chrS", n)], recs)
print(f"wrote {len(recs)} reads, {os.path.getsize(bam)/1e6:.0f} MB compressed in {time.time()-t0:.0f} s", flush=True)
h = api.RsiHot(0)
for rep in range(3):
    st = h.load_depth_bam(bam, "chrS")
    print(f"loader: {st['t_total_ms']:.0f} ms total ({st['bytes_compressed']/1e6:.0f} MB compressed -> {st['bytes_inflated']/1e6:.0f} MB, "
          f"{st['records']} reads, {st['runs']} runs; inflate {st['t_inflate_ms']:.0f} ms, walk {st['t_walk_ms']:.0f} ms, device wait {st['t_wait_ms']:.0f} ms) = {n/st['t_total_ms']/1e3:.1f} Mbases/s of chromosome, "
          f"{st['records']/st['t_total_ms']/1e3:.2f} M reads/s, {st['bytes_compressed']/st['t_total_ms']/1e6:.2f} GB/s compressed", flush=True)
rd = h.fetch("depth_in")
print("mean depth", rd.mean())
os.remove(bam); os.rmdir(d)
