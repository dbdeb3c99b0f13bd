#!/usr/bin/env python3
"""bench.py -- bases/s through the RSI read-depth hot path on MI355X.

A "step" is one pass of the whole hot path (GC correction -> cap -> N removal -> bins -> NB
transform -> RSI scan -> calls) over one synthetic genome already resident in HBM: the 24
chromosomes totalling 3.0 Gb at 30x of BASELINE.json configs[3] with `-m 101 -NB` (the workload
north_star's target is quoted on: ">= 50 Mbases/s ... on a 3 Gb synthetic genome at 1 MI355X").
With N > 1 every rank (one process per GPU) processes its own genome (different sample seed);
the per-chromosome summaries (chr median/SD + calls) are all-gathered over RCCL once per step
-- the path's only exchange -- so scaling is weak and `value` is the whole-job bases/s.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see README "Benchmark").  `roofline` is for the dominant kernel,
timed with HIP events on the library's own stream inside the timed region; `cpu_baseline` times
the compiled reference (oracle/_ref, kind "reference") or the CPU restatement (kind "port") on a
bounded sample of the same workload, on one host core.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM bytes per base of the per-base kernels
ALGO_BYTES_PER_BASE = {"gc_hist": 5.0, "gc_rescale": 9.0, "cap_compact_bin": 9.1, "fasta_classify": 1.0,
                       "value_hist": 4.0}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
MAX_CALLS = 256         # per-chromosome slots in the gathered result block
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # written by tools/pmc_summary.py --json


def pmc_traffic_per_base(kernel):
    """HBM bytes per base of a streaming kernel from the committed rocprofv3 PMC passes (FETCH_SIZE
    and WRITE_SIZE collected in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md section
    HBM prescribes for wide coalesced reads on gfx950).  None when no measurement is committed."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        k = d["kernels"][kernel]
        return (2.0 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0 / d["bases_per_launch"]
    except Exception:
        return None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json config (1-based as in SURVEY 8d): 2, 3, 4 or 5")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink chromosome lengths (debug only; invalidates the metric)")
    ap.add_argument("--cpu-sample-mb", type=float, default=60.0, help="size of the CPU-baseline sample chromosome")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the side measurement of the single-chromosome configs")
    ap.add_argument("--workers", type=int, default=12, help="host threads / HIP streams per GPU (chromosomes in flight)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from rsicnv_amd import api, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # RSI_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices, the
    # collectives run on host tensors); the driver's runs use nccl (= RCCL) with one GPU per rank.
    backend = os.environ.get("RSI_BENCH_BACKEND", "nccl")
    cpu_slot = local_rank
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        # One slice of the allowed CPUs per rank (contiguous ids are normally one socket): a rank's dozen worker threads
        # and the runtime's helper threads then stay next to each other instead of wandering over both sockets.  Only
        # when the slice is comfortably larger than the pool; RSI_BENCH_PIN=0 turns it off.
        try:
            allowed = sorted(os.sched_getaffinity(0))
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
            per = len(allowed) // max(local_world, 1)
            if os.environ.get("RSI_BENCH_PIN", "1") != "0" and per >= args.workers + 4:
                os.sched_setaffinity(0, set(allowed[cpu_slot * per:(cpu_slot + 1) * per]))
        except (AttributeError, OSError):
            pass

    lib = api.load_library()
    pool = api.RsiPool(local_rank, args.workers)
    # HIP events around the per-base (HBM-bound) kernels only: those are the roofline's kernels.  Event pairs around all
    # sixty launches per chromosome cost 13 % of the step; the full per-kernel table comes from one extra, untimed pass.
    pool.set_timing(2)
    flags = synth.config_flags(args.config)
    params = api.make_params(**flags)

    # ---- synthetic genome, generated directly in HBM (not timed) ----
    chroms = [0] if args.config in (2, 3) else list(range(24))
    plans = []
    for c in chroms:
        p = synth.config_plan(args.config, chrom=c, scale=args.scale)
        p["seed"] = p["seed"] + 1000003 * rank     # every rank = another sample of the cohort
        plans.append(p)
    t0 = time.time()
    dev = torch.device("cuda", local_rank)
    data = []
    for p in plans:
        d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device=dev)
        d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device=dev)
        synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
        data.append((d_rd, d_fa, p["n"]))
    torch.cuda.synchronize()
    total_bases = sum(n for _, _, n in data)
    if rank == 0:
        log(f"[bench] generated {len(data)} chromosomes, {total_bases/1e9:.3f} Gb per rank in {time.time()-t0:.1f} s")

    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    gather_in = torch.zeros(len(data), 4 + 4 * MAX_CALLS, dtype=torch.float64, device=coll_dev)
    gather_out = [torch.zeros_like(gather_in) for _ in range(world)] if world > 1 else None

    chrom_args = [(d_rd.data_ptr(), d_fa.data_ptr(), n) for d_rd, d_fa, n in data]

    def step(timed=False):
        results = pool.run(params, chrom_args, collect_times=timed)
        host_block = np.zeros((len(data), 4 + 4 * MAX_CALLS), dtype=np.float64)
        for ci, res in enumerate(results):   # per chromosome: median, SD, number of calls, calls (start, end, type, qscore)
            res.summary_into(host_block[ci], MAX_CALLS)
        if world > 1:   # the one exchange of the path: per-chromosome summaries to every rank
            gather_in.copy_(torch.from_numpy(host_block))
            dist.all_gather(gather_out, gather_in)
        return int(host_block[:, 2].sum())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    pool.reset_times()
    fence()
    t_start = time.perf_counter()
    ncalls = 0
    for _ in range(args.steps):
        ncalls = step(timed=True)
    fence()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (HIP events on the library's streams, timed region) ----
    per_kernel = pool.kernel_table()      # name -> (sum ms, launches, sum of chromosome lengths)

    def kernel_roofline(table, name):
        ms, cnt, bases = table[name]
        byts = ALGO_BYTES_PER_BASE[name] * bases
        achieved = byts / (ms * 1e-3) / 1e9
        return ms, cnt, bases, byts, achieved

    roofline = None
    streaming = [k for k in per_kernel if k in ALGO_BYTES_PER_BASE]
    if streaming:
        dom = max(streaming, key=lambda k: per_kernel[k][0])
        ms, cnt, bases, byts, achieved = kernel_roofline(per_kernel, dom)
        tpb = pmc_traffic_per_base(dom)
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None if tpb is None else round(tpb * bases / cnt),
                    "avg_launch_ms": round(ms / cnt, 4), "launches": int(cnt),
                    "algorithmic_bytes_per_base": ALGO_BYTES_PER_BASE[dom],
                    "algorithmic_bytes_per_launch": round(byts / cnt),
                    "measured": "in situ: the timed steps, with the bin-level kernels of up to "
                                f"{args.workers - 1} other chromosomes and a second per-base phase sharing the GPU"}
        # The same kernel with the per-base phases run alone on the chip (rsi_pool_set_schedule isolate=1):
        # one extra, untimed genome pass; `value` above does not include it.
        pool.set_schedule(isolate=True)
        timed_tables = (pool.times, )
        pool.reset_times()
        fence()
        step(timed=True)
        fence()
        iso = pool.kernel_table()
        pool.set_schedule(isolate=False)
        if dom in iso:
            ims, icnt, ibases, ibyts, iach = kernel_roofline(iso, dom)
            roofline["isolated"] = {"achieved": round(iach, 1), "frac": round(iach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ims / icnt, 4),
                                    "launches": int(icnt), "note": "same launches with the chip to themselves (untimed extra pass)"}
        pool.times = timed_tables[0]
    kernel_ms = {k: round(v[0] / args.steps, 3) for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][0])}
    # every launch of one more untimed pass (normal schedule), for the per-kernel picture of the whole path
    timed_table = pool.times
    pool.reset_times()
    pool.set_timing(1)
    fence()
    step(timed=True)
    fence()
    kernel_ms_all = {k: round(v[0], 3) for k, v in sorted(pool.kernel_table().items(), key=lambda kv: -kv[1][0])}
    pool.set_timing(2)
    pool.times = timed_table
    phase_ms = pool.phase_table()

    # ---- the single-chromosome configurations (configs[1], configs[2]) on the same pool: latency of ONE chromosome, i.e.
    # no other chromosome to overlap with; reported next to `value`, never part of it ----
    single = None
    if rank == 0 and world == 1 and args.config == 4 and args.scale == 1.0 and not args.no_single:
        single = {}
        pool.set_timing(0)
        for cfg, label in ((2, "configs[1]: one 60 Mb chromosome, 30x Poisson"), (3, "configs[2]: one 250 Mb chromosome, 30x gamma-Poisson + GC")):
            sp = synth.config_plan(cfg)
            s_fa = torch.empty(sp["n"] + 64, dtype=torch.uint8, device=dev)
            s_rd = torch.empty(sp["n"] + 16, dtype=torch.int32, device=dev)
            synth.generate_device(lib, sp, s_fa.data_ptr(), s_rd.data_ptr())
            torch.cuda.synchronize()
            sparams = api.make_params(**synth.config_flags(cfg))
            one = [(s_rd.data_ptr(), s_fa.data_ptr(), sp["n"])]
            for _ in range(2):
                pool.run(sparams, one)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                r1 = pool.run(sparams, one)
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t1) / reps
            single[label] = {"ms": round(dt1 * 1e3, 3), "bases_per_s": round(sp["n"] / dt1, 1), "calls": len(r1[0].calls("calls"))}
            del s_fa, s_rd
        pool.set_timing(2)

    # ---- CPU baseline on a bounded sample (rank 0, N = 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(lib, args, flags)

    if rank == 0:
        value = world * total_bases * args.steps / elapsed
        out = {
            "metric": "bases/sec through RSI pipeline (bin+GC+NB+segment)", "value": round(value, 1), "unit": "bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload_name(args), "chromosomes": len(data), "bases_per_gpu": total_bases,
                       "flags": flag_string(flags), "calls_per_genome": ncalls,
                       "parallelism": f"{world} rank(s), one genome per GPU, {args.workers} chromosomes in flight per GPU, "
                                      "all_gather of per-chromosome summaries"},
            "roofline": roofline, "cpu_baseline": cpu, "kernel_ms_per_step": kernel_ms,
            "kernel_ms_all_launches_extra_pass": kernel_ms_all, "single_chromosome_configs": single,
            "worker_phase_ms_per_step": {k: round(v / args.steps, 2) for k, v in sorted(phase_ms.items(), key=lambda kv: -kv[1])},
        }
        print(json.dumps(out), flush=True)
    pool.close()
    if world > 1:
        dist.destroy_process_group()


def flag_string(f):
    return f"-m {f['m']} " + ("-NB" if f["trans"] == 0 else "-MED" if f["trans"] == 1 else "-ALL") + f" -cap {f['cap']:g}" + \
        ("" if f["gcadjust"] else " -NOGC")


def workload_name(args):
    names = {2: "synthetic 60 Mb chromosome, 30x Poisson depth (configs[1])",
             3: "synthetic 250 Mb chromosome, 30x gamma-Poisson depth with GC dependence (configs[2])",
             4: "24 synthetic chromosomes totalling 3 Gb, 30x, per GPU (configs[3] data; north_star 3 Gb genome)",
             5: "24 synthetic chromosomes totalling 3 Gb, 60x, -m 51 -MED -cap 4, per GPU (configs[4] data)"}
    s = names[args.config]
    if args.scale != 1.0:
        s += f" [scaled x{args.scale}: NOT the metric's configuration]"
    return s


def cpu_baseline(lib, args, flags):
    """Reference (or the oracle port) on one chromosome of the same model, compute-only, one core."""
    import oracle
    from rsicnv_amd import synth
    n = int(args.cpu_sample_mb * 1e6 * min(args.scale, 1.0)) if args.scale < 1 else int(args.cpu_sample_mb * 1e6)
    model = 0 if args.config == 2 else 1
    mean = 60.0 if args.config == 5 else 30.0
    plan = synth.make_plan(n, 0xC0FFEE, model=model, mean=mean, n_events=20 if model else 9, gaps=2,
                           centromere=int(1_000_000 * min(args.scale, 1.0)) if model else 0)
    fasta, depth = synth.generate_host(lib, plan)
    p = oracle.make_params(**flags)
    t0 = time.perf_counter()
    if oracle.ref_available():
        nc, stages = oracle.Ref().run_timed(p, depth, fasta)
        kind = "reference"
    else:
        if not os.path.exists(oracle.ORACLE_SO):
            import subprocess
            subprocess.run(["make", "-f", "oracle/Makefile", "oracle/librsi_oracle.so"], cwd=ROOT, check=True)
        O = oracle.Oracle()
        nc = O.run(p, depth, fasta, snapshots=False)
        stages = list(O.f64("stage_s"))
        kind = "port"
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "bases/s", "cores": 1, "kind": kind,
            "sample": f"one {n/1e6:.0f} Mb chromosome of the same depth model and flags, compute-only "
                      f"(arrays in memory -> calls), {dt:.1f} s, {nc} calls",
            "stage_s": [round(s, 3) for s in stages]}


if __name__ == "__main__":
    main()
