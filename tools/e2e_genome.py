"""End to end on the 3 Gb genome, where north_star states its target (">= 50 Mbases/s end-to-end ... on a 3 Gb synthetic genome"):
the 24 chromosomes of configs[3] as FASTA + "pos depth" text files through the command line, `rsicnv rsi -f REF -d RDFILE -c CHR -o OUT`
(process start, FASTA read, depth text parse, device path, output file: loaddata.cpp:473-539, rsi.cpp:2069-2259) -- one after the
other, and four at a time (a GPU box allows few processes on its card).  The files need 12 bytes per base; when the scratch
directory cannot hold the whole genome the largest prefix of chromosomes (longest first) that fits is run, and the record says so.

usage: e2e_genome.py [--dir SCRATCH] [--out JSON] [--config 4] [--parallel 4] [--max-gb G]"""
import argparse, json, os, shutil, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rsicnv_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--dir", default="/tmp/rsi_e2e")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r5_e2e_genome.json"))
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--parallel", type=int, default=4)
ap.add_argument("--max-gb", type=float, default=0.0, help="cap on the bytes of files (0: what the scratch directory has free, minus 4 GB)")
args = ap.parse_args()

lib = api.load_library()
os.makedirs(args.dir, exist_ok=True)
free = shutil.disk_usage(args.dir).free
budget = (args.max_gb * (1 << 30)) if args.max_gb > 0 else max(0, free - 4 * (1 << 30))
plans = [synth.config_plan(args.config, chrom=c) for c in range(24)]
order = sorted(range(24), key=lambda c: -plans[c]["n"])
chosen, need = [], 0
for c in order:
    b = int(plans[c]["n"] * 13.2)   # ~11.8 bytes of text per base + 1 of FASTA, with margin
    if need + b <= budget:
        chosen.append(c); need += b
if not chosen:
    raise SystemExit(f"e2e_genome: {free / 1e9:.1f} GB free under {args.dir}: not even the smallest chromosome fits")
flags = synth.config_flags(args.config)
exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
flag_args = ["-m", str(flags["m"])] + (["-MED"] if flags.get("trans", 0) == 1 else []) + (["-cap", str(flags["cap"])] if "cap" in flags else [])
torch.cuda.set_device(0)
t0 = time.time()
cases = []
for c in chosen:
    p = plans[c]
    d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda"); d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
    torch.cuda.synchronize()
    fasta = d_fa[:p["n"]].cpu().numpy(); depth = d_rd[:p["n"]].cpu().numpy()
    del d_fa, d_rd
    sub = os.path.join(args.dir, f"chr{c + 1}")
    os.makedirs(sub, exist_ok=True)
    fa, rdf = synth.write_case_files(lib, fasta, depth, sub, chrom=f"chr{c + 1}")
    cases.append((c, p["n"], fa, rdf, os.path.join(sub, "out.txt"), os.path.getsize(rdf)))
    del fasta, depth
torch.cuda.empty_cache()
t_files = time.time() - t0
bases = sum(n for _, n, *_ in cases)
text_bytes = sum(b for *_, b in cases)
print(f"[e2e] {len(cases)} of 24 chromosomes, {bases / 1e9:.3f} Gb, {text_bytes / 1e9:.1f} GB of depth text written in {t_files:.0f} s", flush=True)

def cmd(c, fa, rdf, out):
    return [exe, "rsi", "-f", fa, "-d", rdf, "-c", f"chr{c + 1}", "-o", out, "-np"] + flag_args

def run_all(par):
    t = time.perf_counter()
    if par <= 1:
        for c, n, fa, rdf, out, _ in cases:
            r = subprocess.run(cmd(c, fa, rdf, out), capture_output=True, timeout=1200)
            if r.returncode != 0:
                raise RuntimeError(f"chr{c + 1}: " + r.stderr.decode()[-300:])
    else:
        pending = list(cases); running = []
        while pending or running:
            while pending and len(running) < par:
                c, n, fa, rdf, out, _ = pending.pop(0)
                running.append((c, subprocess.Popen(cmd(c, fa, rdf, out), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
            for item in list(running):
                c, pr = item
                if pr.poll() is not None:
                    if pr.returncode != 0:
                        raise RuntimeError(f"chr{c + 1}: " + pr.stderr.read().decode()[-300:])
                    running.remove(item)
            time.sleep(0.002)
    return time.perf_counter() - t

run_all(1)                      # warm: page cache, the device's first allocations
t_seq = run_all(1)
t_par = run_all(args.parallel)
calls = sum(sum(1 for l in open(out) if not l.startswith("#")) for *_, out, _ in cases)
rec = {"config": f"configs[{args.config - 1}]: {flags}", "chromosomes_run": len(cases), "chromosomes_of_genome": 24, "bases": bases,
       "whole_genome": len(cases) == 24, "depth_text_bytes": text_bytes, "scratch_free_bytes_at_start": free, "files_written_in_s": round(t_files, 1),
       "sequential": {"s": round(t_seq, 3), "bases_per_s": round(bases / t_seq, 1)},
       f"{args.parallel}_processes_at_once": {"s": round(t_par, 3), "bases_per_s": round(bases / t_par, 1)},
       "calls": calls, "north_star_target_bases_per_s": 50e6,
       "note": "rsicnv rsi -f REF -d RDFILE -c CHR -o OUT per chromosome: process start, FASTA, depth text parse (files in the page cache as far as it holds them), device path, output file"}
with open(args.out, "w") as f:
    json.dump(rec, f, indent=1)
print(json.dumps(rec), flush=True)
shutil.rmtree(args.dir, ignore_errors=True)
