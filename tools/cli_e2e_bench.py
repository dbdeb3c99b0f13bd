"""End-to-end wall time of the command line on one chromosome (t_e2e, parse included).
Usage: python tools/cli_e2e_bench.py [Mb=60]"""
import os, sys, time, tempfile, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rsicnv_amd import api, synth

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = api.load_library()
plan = synth.config_plan(2, chrom=0, scale=mb / 60.0)
fasta, depth = synth.generate_host(lib, plan)
n = depth.size
d = tempfile.mkdtemp(dir="/tmp")
fa = os.path.join(d, "ref.fa")
with open(fa, "wb") as f:
    f.write(b">chrS\n")
    seq = fasta.tobytes()
    f.write(b"\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + b"\n")
open(fa + ".fai", "w").write(f"chrS\t{n}\t6\t60\t61\n")
rd = os.path.join(d, "depth.txt")
pos = np.arange(1, n + 1)
with open(rd, "w") as f:
    step = 2_000_000
    for a in range(0, n, step):
        b = min(n, a + step)
        f.write("\n".join(f"{p}\t{v}" for p, v in zip(pos[a:b].tolist(), depth[a:b].tolist())) + "\n")
print(f"{n/1e6:.0f} Mb: fasta {os.path.getsize(fa)/1e6:.0f} MB, depth text {os.path.getsize(rd)/1e6:.0f} MB", flush=True)
exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
for rep in range(3):
    t0 = time.time()
    r = subprocess.run([exe, "rsi", "-f", fa, "-d", rd, "-c", "chrS", "-o", os.path.join(d, "out.txt"), "-np"], capture_output=True)
    dt = time.time() - t0
    tim = [l for l in r.stderr.decode().splitlines() if l.startswith("timing")]
    print(f"run {rep}: wall {dt:.2f} s = {n/dt/1e6:.0f} Mbases/s end to end | {tim[0] if tim else ''}", flush=True)
print(open(os.path.join(d, "out.txt")).read()[:600])
