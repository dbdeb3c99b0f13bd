#!/usr/bin/env python3
"""bench.py -- bases/s through the RSI read-depth hot path on MI355X.

A "step" is one pass of the whole hot path (GC correction -> cap -> N removal -> bins -> NB
transform -> RSI scan -> calls -> rows in chromosome order) over ONE synthetic genome already
resident in HBM: the 24 chromosomes totalling 3.0 Gb at 30x of BASELINE.json configs[3] with
`-m 101 -NB` (the workload north_star's target is quoted on: ">= 50 Mbases/s ... on a 3 Gb synthetic
genome at 1 MI355X").

With N > 1 (one process per GPU) the genome is SHARDED by chromosome, longest first, each to the
least loaded rank (configs[3]: "sharded per-chrom across 8 MI355X"): the parallel form of the
reference's loop rsi.cpp:2189-2217.  Ranks never exchange depth data; the one collective per step
is an all_gather of the per-chromosome summary blocks (chromosome median / SD + calls, rsi_result_
summary) over RCCL, after which rank 0 puts the rows in chromosome order as the reference's writer
does (rsi.cpp:1594-1608).  Total work is fixed as N grows: `scaling` is "strong" and `value` is
genome bases x steps / time.  `--shard sample` is the other reading (every rank its own genome of
a cohort, weak scaling), reported with that label.

  python bench.py --gpus 1 --steps 8 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Steps are SUBMITTED to the pool (rsi_pool_submit) and waited for, gathered and turned into rows on a second host thread:
`config.steps_in_flight` genomes are queued at once (2 for a whole genome on one GPU, as many as fill the pool's workers for
a rank's few chromosomes; `--inflight 1` = one genome at a time), the way samples arrive in production -- the first
chromosomes of genome k + 1 run beside the last ones of genome k.  Every step's rows are produced and hashed inside the timed
region; the timer is read after the last step's rows exist (barrier + synchronize on both sides, max over ranks).

Rank 0 prints ONE JSON line.  In it:
  roofline      the dominant per-base kernel -- the one an untimed pass with HIP events around every
                per-base launch MEASURES as the longest -- then timed inside the timed steps (events
                around that kernel only, on the library's own streams): algorithmic bytes (SURVEY 8d's
                row of the kernel) over that time; `isolated` = the same launches with the chip to
                themselves (one extra, untimed pass); `streaming_kernels` = every per-base kernel with
                three fractions each, in situ and isolated: `algorithmic` (SURVEY's bytes), `actual`
                (the bytes the committed PMC passes counted, profiles/pmc_traffic.json) and `valu` (the
                vector instructions the PMC passes counted over the chip's 1.23e12 wave-instructions/s);
                `whole_path` = 23.7 (m = 101) / 24.3 (m = 51) algorithmic B/base x genome bases over
                the step time: the fraction BASELINE.md defines for the path; `scan` = window
                evaluations per second of the RSI scan kernel (2 sweeps x Lmax x bins x 2 passes).
  steps_identical  every timed step's rows hashed: true when all steps produced the same bytes.
  cpu_baseline  the compiled reference (oracle/_ref, kind "reference") or the CPU restatement
                (kind "port") on one 60 Mb chromosome of the same model, one core; `all_cores` =
                one such process per host core at once.
  t_device_h2d  rsi_hot_run from pinned host buffers (H2D of the inputs included) for configs[1]
                and configs[2]; t_e2e = the command line on a 60 Mb depth text file (parse included).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM bytes per base of the per-base kernels
# (K4 is two launches since round 5: `cap_compact_bin` = K4s, SURVEY's row P3 -- read 4 + 1, write 4 per base, + 12 per bin -- and
# `bin_median` = K4m, which an ideally fused pipeline would not need: it is charged the byte per base it reads plus its 12 bytes per bin)
ALGO_BYTES_PER_BASE = {"gc_hist": 5.0, "gc_joint_hist": 5.0, "value_hist8": 9.0, "gc_rescale": 9.0, "cap_compact_bin": 9.1,
                       "fasta_classify": 1.0, "value_hist": 4.0, "bin_median": 1.12}
WHOLE_PATH_BYTES_PER_BASE = {101: 23.7, 51: 24.3}   # SURVEY 8d, GC on
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 2   # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz = 1.23e12 /s
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


def pmc_valu_per_base(kernel):
    """Vector instructions (wave-instructions, SQ_INSTS_VALU) per base of a kernel from the committed PMC passes."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        return d["kernels"][kernel]["SQ_INSTS_VALU"] / d["bases_per_launch"]
    except Exception:
        return None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json config (1-based as in SURVEY 8d): 2, 3, 4 or 5")
    ap.add_argument("--shard", choices=("genome", "sample"), default="genome",
                    help="N > 1: one genome sharded by chromosome over the ranks (strong scaling, the north_star mode) or one genome per rank (weak)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink chromosome lengths (debug only; invalidates the metric)")
    ap.add_argument("--cpu-sample-mb", type=float, default=60.0, help="size of the CPU-baseline sample chromosome")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the side measurements (single chromosomes, host-buffer runs, command line)")
    ap.add_argument("--workers", type=int, default=0, help="host threads / HIP streams per GPU (chromosomes in flight); 0 = min(20 on one GPU / 16 per rank of several, max(4, 2 x the "
                    "host cores this rank may use)): sixteen on a box with eight cores and more per rank, fewer where eight ranks share sixteen cores")
    ap.add_argument("--inflight", type=int, default=0, help="steps (genomes) queued in the pool at once (rsi_pool_submit): the next genome's first "
                    "chromosomes run beside the last ones of the current genome.  0 = as many as keep the pool's workers busy with this rank's "
                    "share (2 for a whole genome, more for the few chromosomes of one rank among eight, at most 6); 1 = one genome at a time")
    ap.add_argument("--check-rows", choices=("auto", "off"), default="auto",
                    help="auto: a full-size configs[3] / configs[4] genome's rows must hash to what the reference wrote (tests/golden/genome_rows.json)")
    args = ap.parse_args()

    # ---- `--gpus N` without a launcher: this process -- before torch is imported or a GPU touched -- starts the N ranks itself
    # (one process per GPU, torch.distributed.run on 127.0.0.1), relays rank 0's line and exits with the launcher's code.  A
    # line with n_gpus != --gpus cannot come out of this file: a WORLD_SIZE that disagrees with --gpus is an error. ----
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # --standalone: the launcher picks a free rendezvous port itself (binding one here and closing it again left a window in
        # which somebody else could take it); --local-addr: the container's hostname may not resolve
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               os.path.abspath(__file__)] + sys.argv[1:]
        log("[bench] no launcher in the environment: starting", " ".join(cmd))
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to print a line for the wrong number of GPUs")

    import numpy as np
    import torch
    import torch.distributed as dist
    from rsicnv_amd import api, synth
    from rsicnv_amd import dist as rd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # RSI_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices, the
    # collectives run on host tensors); the driver's runs use nccl (= RCCL) with one GPU per rank.
    backend = os.environ.get("RSI_BENCH_BACKEND", "nccl")
    cpu_slot = local_rank
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # The host cores this rank may use: its share of the process's CPU set (RSI_BENCH_CPUS_PER_RANK=k: pretend the box grants k per
    # rank -- the rehearsal of eight ranks on a sixteen-core host).  The pool's size follows it unless --workers says otherwise.
    cores_per_rank = None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        cores_per_rank = max(1, len(allowed) // max(local_world, 1))
        forced = int(os.environ.get("RSI_BENCH_CPUS_PER_RANK", "0"))
        if forced > 0:
            cores_per_rank = min(cores_per_rank, forced)
    except (AttributeError, OSError):
        allowed, forced = None, 0
    if args.workers <= 0:
        # 20 on one GPU; 16 per rank when there are several: a process gets 24 hardware queues, the 22nd worker stream already shares one
        # with something else (21 workers 11.7 ms per genome, 22 workers 14-17, tools/gpu_env_sweep.sh), and RCCL brings streams of its own
        top = 20 if world == 1 else 16
        args.workers = top if cores_per_rank is None else min(top, max(4, 2 * cores_per_rank))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        # One slice of the allowed CPUs per rank (contiguous ids are normally one socket): a rank's dozen worker threads
        # and the runtime's helper threads then stay next to each other instead of wandering over both sockets.  Only
        # when the slice is comfortably larger than the pool (or the slice was asked for); RSI_BENCH_PIN=0 turns it off.
        try:
            if allowed is not None and os.environ.get("RSI_BENCH_PIN", "1") != "0" and (forced > 0 or cores_per_rank >= args.workers + 4):
                os.sched_setaffinity(0, set(allowed[cpu_slot * cores_per_rank:(cpu_slot + 1) * cores_per_rank]))
        except (AttributeError, OSError):
            pass
    elif allowed is not None and forced > 0:
        os.sched_setaffinity(0, set(allowed[:cores_per_rank]))

    lib = api.load_library()
    pool = api.RsiPool(local_rank, args.workers)
    # HIP events around the dominant per-base kernel only inside the timed steps (event records are not free); the
    # per-kernel tables come from extra, untimed passes.
    pool.set_timing(0)
    flags = synth.config_flags(args.config)
    params = api.make_params(**flags)

    # ---- the genome: every chromosome's plan; this rank generates (directly in HBM, not timed) the ones it will process ----
    chrom_ids = [0] if args.config in (2, 3) else list(range(24))
    sharded = args.shard == "genome"
    plans = []
    for c in chrom_ids:
        p = synth.config_plan(args.config, chrom=c, scale=args.scale)
        if not sharded:
            p["seed"] = p["seed"] + 1000003 * rank     # every rank = another sample of the cohort
        plans.append(p)
    lengths = [p["n"] for p in plans]
    parts = rd.lpt_assign(lengths, world) if sharded else [list(range(len(plans)))] * world
    mine = parts[rank]
    t0 = time.time()
    dev = torch.device("cuda", local_rank)
    data = {}
    for i in mine:
        p = plans[i]
        d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device=dev)
        d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device=dev)
        synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
        data[i] = (d_rd, d_fa, p["n"])
    torch.cuda.synchronize()
    my_bases = sum(data[i][2] for i in mine)
    genome_bases = sum(lengths)
    if rank == 0:
        log(f"[bench] rank 0: {len(mine)} of {len(plans)} chromosomes, {my_bases/1e9:.3f} of {genome_bases/1e9:.3f} Gb, generated in {time.time()-t0:.1f} s")

    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    nslots = max(len(p) for p in parts)
    names = [f"chr{c + 1}" for c in chrom_ids]
    chrom_args = [(data[i][0].data_ptr(), data[i][1].data_ptr(), data[i][2]) for i in mine]

    import hashlib
    step_hashes = []
    scan_evals = [0]
    id_offset = 0 if sharded else rank * len(plans)   # weak-scaling mode: every rank's sample keeps its own ids

    # A step = the pool's pass over this rank's chromosomes, then `finish_step` (pack, gather, rank 0's rows in order, hash).
    # A step is SUBMITTED to the pool (rsi_pool_submit) and waited for on a second host thread, which then runs finish_step:
    # up to --inflight steps are queued in the pool at once, so the first chromosomes of genome k + 1 run beside the last ones
    # of genome k (samples back to back, the way a sequencing centre runs them), and the GPU never waits for Python.  Every
    # step's rows are still produced (and hashed) inside the timed region -- `drain()` joins the thread before a timer is read.
    # One thread, one queue: the collectives stay in step order on every rank.
    import queue
    import threading
    post_q = queue.Queue()
    post_err = []
    last_rows = [0]
    if args.inflight <= 0:   # enough queued genomes that every worker has a chromosome of this rank's share
        args.inflight = max(2, min(6, (args.workers + max(1, len(mine)) - 1) // max(1, len(mine))))
    slots = threading.Semaphore(max(1, args.inflight))

    def post_worker():
        torch.cuda.set_device(dev)   # a new thread starts on device 0: the gather's tensors and RCCL's streams belong to this rank's GPU
        while True:
            item = post_q.get()
            try:
                if item is None:
                    return
                handle, timed = item
                results = pool.wait(handle) if handle is not None else []
                if not post_err:
                    last_rows[0] = finish_step(results, timed)
            except Exception as e:   # surfaces at the next drain()
                post_err.append(e)
            finally:
                slots.release()
                post_q.task_done()

    post_thread = threading.Thread(target=post_worker, daemon=True)
    post_thread.start()

    def drain():
        post_q.join()
        if post_err:
            raise post_err[0]
        return last_rows[0]

    def step(timed=False):
        """One genome: this rank's chromosomes through the pool; the rest of the step is queued for the second thread."""
        slots.acquire()   # at most --inflight steps between submission and their rows
        handle = pool.submit(params, chrom_args, collect_times=timed) if chrom_args else None
        post_q.put((handle, timed))

    def finish_step(results, timed):
        """The gather, rank 0's rows in chromosome order."""
        if timed:   # window evaluations of the scan this step: 2 sweeps x Lmax x bins x 2 passes per chromosome
            scan_evals[0] = sum(4 * int(r.stats["Lmax"]) * int(r.stats["nbins"]) for r in results)
        block = rd.pack_results(mine, results, nslots, id_offset)
        if world > 1:   # the one exchange of the path
            blocks = rd.gather_blocks(block, world, coll_dev)
        else:
            blocks = [block]
        if rank != 0:
            return 0
        merged = rd.unpack_blocks(blocks)
        if len(merged) != (len(plans) if sharded else world * len(plans)):
            raise RuntimeError(f"gather returned {len(merged)} chromosomes of {len(plans)} x {1 if sharded else world}")
        rows = rd.format_rows(lib, merged, names if sharded else [names[c % len(plans)] for c in range(world * len(plans))])
        if timed:   # a step whose rows differ from another step's is a wrong step: every timed step is hashed
            h = hashlib.sha256()
            for r in rows:
                h.update(r.encode()); h.update(b"\n")
            for c in sorted(merged):
                h.update(repr((c, merged[c]["RDmedian"], merged[c]["RDsd"])).encode())
            step_hashes.append(h.hexdigest())
        return len(rows)

    def fence():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ---- which per-base kernel is the dominant one?  Measured, not assumed: one untimed pass with HIP events around every
    # per-base launch (timing mode 2); the timed steps then bracket that kernel only (event records are not free). ----
    fence()                # the warm-up genomes are done before the timing mode changes under the workers
    pool.reset_times()
    pool.set_timing(2)
    step(timed=True)
    fence()
    insitu_all = pool.kernel_table()
    streaming_names = [k for k in insitu_all if k in ALGO_BYTES_PER_BASE]
    dom = max(streaming_names, key=lambda k: insitu_all[k][0]) if streaming_names else "cap_compact_bin"
    if world > 1:   # every rank brackets the kernel rank 0 measured
        obj = [dom]
        dist.broadcast_object_list(obj, src=0)
        dom = obj[0]
    pool.set_timing_kernel(dom)
    pool.set_timing(3)
    step_hashes.clear()
    pool.reset_times()
    for _ in range(2):     # two untimed steps in the timed steps' own mode
        step()
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    fence()                                   # joins the second thread: all K steps' rows exist
    elapsed = time.perf_counter() - t_start
    ncalls = last_rows[0]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_bases = genome_bases if sharded else world * genome_bases
    ms_per_step = elapsed / args.steps * 1e3
    headline_hashes = list(step_hashes[:args.steps])

    # ---- beside the headline: the same genome ONE AT A TIME (a step is submitted when the previous step's rows exist) ----
    one_at_a_time = None
    if args.inflight > 1:
        saved_slots = slots
        slots = threading.Semaphore(1)
        k1 = max(3, min(args.steps, 10))
        step(); fence()
        t1 = time.perf_counter()
        for _ in range(k1):
            step()
        fence()
        e1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([e1], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e1 = float(t.item())
        one_at_a_time = {"ms_per_step": round(e1 / k1 * 1e3, 3), "steps": k1, "bases_per_s": round(total_bases * k1 / e1, 1),
                         "note": "--inflight 1: no genome is submitted before the previous genome's rows exist"}
        slots = saved_slots

    # ---- roofline of the dominant kernel (HIP events on the library's streams, timed region) ----
    per_kernel = pool.kernel_table()      # name -> (sum ms, launches, sum of chromosome lengths)

    def kernel_roofline(table, name):
        ms, cnt, bases = table[name]
        byts = ALGO_BYTES_PER_BASE[name] * bases
        achieved = byts / (ms * 1e-3) / 1e9
        return ms, cnt, bases, byts, achieved

    def three_fractions(table, name):
        """SURVEY-algorithmic, actual (PMC bytes) and VALU-issue fractions of one kernel's launches in `table`."""
        ms, cnt, bases = table[name]
        sec = ms * 1e-3
        out = {"avg_launch_ms": round(ms / cnt, 4), "launches": int(cnt),
               "algorithmic": round(ALGO_BYTES_PER_BASE[name] * bases / sec / 1e9 / HBM_PEAK_GBS, 4)}
        tpb, vpb = pmc_traffic_per_base(name), pmc_valu_per_base(name)
        out["actual"] = None if tpb is None else round(tpb * bases / sec / 1e9 / HBM_PEAK_GBS, 4)
        out["valu"] = None if vpb is None else round(vpb * bases / sec / VALU_PEAK_WAVE_INSTS, 4)
        return out

    roofline = None
    streaming = [k for k in per_kernel if k in ALGO_BYTES_PER_BASE]
    steps_identical = len(set(step_hashes)) <= 1
    if streaming:
        dom = max(streaming, key=lambda k: per_kernel[k][0])
        ms, cnt, bases, byts, achieved = kernel_roofline(per_kernel, dom)
        tpb = pmc_traffic_per_base(dom)
        whole_bpb = WHOLE_PATH_BYTES_PER_BASE.get(flags["m"], 23.7)
        whole = whole_bpb * total_bases / world / (ms_per_step * 1e-3) / 1e9   # per GPU
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None if tpb is None else round(tpb * bases / cnt),
                    "avg_launch_ms": round(ms / cnt, 4), "launches": int(cnt),
                    "algorithmic_bytes_per_base": ALGO_BYTES_PER_BASE[dom],
                    "algorithmic_bytes_per_launch": round(byts / cnt),
                    "measured": "in situ: the timed steps, with the kernels of up to "
                                f"{args.workers - 1} other chromosomes sharing the GPU; the kernel is the per-base kernel with the largest "
                                "summed HIP-event time in an untimed pass that bracketed every per-base launch",
                    "dominant_pick_ms": {k: round(insitu_all[k][0], 3) for k in sorted(streaming_names, key=lambda k: -insitu_all[k][0])},
                    "whole_path": {"algorithmic_bytes_per_base": whole_bpb, "achieved": round(whole, 1), "frac": round(whole / HBM_PEAK_GBS, 4),
                                   "note": "SURVEY 8d's whole-path algorithmic bytes x the bases one GPU processed per step / step time"}}
        # The same kernel with the per-base phases run alone on the chip (rsi_pool_set_schedule isolate=1):
        # one extra, untimed genome pass; `value` above does not include it.
        fence()
        pool.set_schedule(isolate=True)
        pool.set_timing(2)
        timed_tables = (pool.times, )
        pool.reset_times()
        step(timed=True)
        fence()
        iso = pool.kernel_table()
        del step_hashes[args.steps:]
        pool.set_schedule(isolate=False)
        pool.set_timing(3)
        if dom in iso:
            ims, icnt, ibases, ibyts, iach = kernel_roofline(iso, dom)
            roofline["isolated"] = {"achieved": round(iach, 1), "frac": round(iach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ims / icnt, 4),
                                    "launches": int(icnt), "note": "same launches with the chip to themselves (untimed extra pass)"}
        # every per-base kernel, three fractions each: against SURVEY's algorithmic bytes, against the bytes the kernel
        # really moves (PMC), against the chip's vector-instruction issue rate (PMC) -- in situ (the untimed pass that picked
        # the dominant kernel) and isolated
        roofline["streaming_kernels"] = {
            k: {"algorithmic_bytes_per_base": ALGO_BYTES_PER_BASE[k],
                "actual_bytes_per_base": None if pmc_traffic_per_base(k) is None else round(pmc_traffic_per_base(k), 3),
                "valu_insts_per_base_x64": None if pmc_valu_per_base(k) is None else round(64 * pmc_valu_per_base(k), 2),
                "in_situ": three_fractions(insitu_all, k), "isolated": three_fractions(iso, k) if k in iso else None}
            for k in sorted(streaming_names, key=lambda k: -insitu_all[k][0])}
        pool.times = timed_tables[0]
    # every launch of one more untimed pass (normal schedule), for the per-kernel picture of the whole path
    fence()
    timed_table = pool.times
    pool.reset_times()
    pool.set_timing(1)
    step(timed=True)
    fence()
    del step_hashes[args.steps:]
    all_table = pool.kernel_table()
    kernel_ms_all = {k: round(v[0], 3) for k, v in sorted(all_table.items(), key=lambda kv: -kv[1][0])}
    if roofline is not None and "rsi_scan" in all_table and all_table["rsi_scan"][0] > 0:
        sms = all_table["rsi_scan"][0]
        roofline["scan"] = {"kernel": "rsi_scan", "bound": "lds/alu", "window_evaluations_per_step": int(scan_evals[0]),
                            "kernel_ms_per_step_in_situ": round(sms, 3),
                            "evaluations_per_s": round(scan_evals[0] / (sms * 1e-3), 1),
                            "note": "2 sweeps x Lmax x bins x 2 passes per chromosome over the kernel's HIP-event time (all launches of one untimed pass)"}
    pool.set_timing(3)
    pool.times = timed_table
    phase_ms = pool.phase_table()

    side = None
    if rank == 0 and world == 1 and args.config == 4 and args.scale == 1.0 and not args.no_single:
        side = side_measurements(lib, pool, dev, args)

    # ---- CPU baseline on a bounded sample (rank 0, N = 1 only) ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(lib, args, flags)

    # ---- who ran where (every rank's device), the collective's backend and library version ----
    dev_name = torch.cuda.get_device_name(local_rank)
    me = {"rank": rank, "device_index": local_rank, "device": dev_name, "chromosomes": len(mine), "bases": my_bases}
    ranks_info = [me]
    if world > 1:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, me)
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl = None

    # ---- the timed steps' rows against the REFERENCE's rows (tests/golden/genome_rows.json, written by
    # tools/make_golden_full.py from the compiled reference's output for the very same generated chromosomes) ----
    rows_ref = None
    rows_match = None
    if args.check_rows == "auto" and sharded and args.scale == 1.0 and args.config in (4, 5):
        try:
            with open(os.path.join(ROOT, "tests", "golden", "genome_rows.json")) as f:
                rows_ref = json.load(f).get(f"config{args.config}")
        except OSError:
            rows_ref = None
        if rows_ref is not None and rank == 0:
            rows_match = bool(headline_hashes) and all(h == rows_ref["rows_sha256"] for h in headline_hashes)

    if rank == 0:
        value = total_bases * args.steps / elapsed
        if sharded:
            par = (f"{world} rank(s), one genome sharded by chromosome (longest first to the least loaded rank): rank 0 runs "
                   f"{len(mine)} of {len(plans)} chromosomes, {args.workers} in flight per GPU; one all_gather of per-chromosome "
                   "summary blocks per step, rank 0 orders the rows")
        else:
            par = f"{world} rank(s), one genome per GPU (another sample each), {args.workers} chromosomes in flight per GPU, all_gather of summaries"
        out = {
            "metric": "bases/sec through RSI pipeline (bin+GC+NB+segment)", "value": round(value, 1), "unit": "bases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload_name(args), "chromosomes": len(plans), "genome_bases": genome_bases,
                       "bases_on_rank0": my_bases, "flags": flag_string(flags), "calls_per_genome": ncalls, "shard": args.shard,
                       "parallelism": par, "steps_in_flight": max(1, args.inflight), "world_size": world, "workers": args.workers,
                       "cores_per_rank": cores_per_rank,
                       "backend": (backend if world > 1 else "none (one rank)"), "rccl_version": rccl, "ranks": ranks_info},
            "one_genome_at_a_time": one_at_a_time,
            "roofline": roofline, "cpu_baseline": cpu,
            "steps_identical": steps_identical, "rows_sha256": headline_hashes[0] if headline_hashes else None,
            "rows_match_reference": rows_match,
            "rows_reference_sha256": None if rows_ref is None else rows_ref["rows_sha256"],
            "kernel_ms_all_launches_extra_pass": kernel_ms_all,
            "worker_phase_ms_per_step": {k: round(v / args.steps, 2) for k, v in sorted(phase_ms.items(), key=lambda kv: -kv[1])},
        }
        if side:
            out.update(side)
        print(json.dumps(out), flush=True)
    drain()
    post_q.put(None)
    post_thread.join(timeout=10)
    pool.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        bad = [w for w, ok in (("headline", rows_match), ("configs[4] side pass", (side or {}).get("configs[4]_side_pass", {}).get("rows_match_reference")))
               if ok is False]
        if bad:
            raise SystemExit(f"bench.py: rows differ from the reference's for: {', '.join(bad)}")


def side_measurements(lib, pool, dev, args):
    """Next to `value`, never part of it: the single-chromosome configurations on resident inputs (latency of ONE chromosome,
    nothing to overlap with), the same from pinned host buffers (t_device_h2d) and the command line on a depth
    text file (t_e2e)."""
    import numpy as np
    import torch
    from rsicnv_amd import api, synth
    out = {"single_chromosome_configs": {}, "t_device_h2d": {}}
    pool.set_timing(0)
    hot = api.RsiHot(dev.index or 0)
    host_case = host_case250 = None
    for cfg, label in ((2, "configs[1]: one 60 Mb chromosome, 30x Poisson"), (3, "configs[2]: one 250 Mb chromosome, 30x gamma-Poisson + GC")):
        sp = synth.config_plan(cfg)
        n = sp["n"]
        s_fa = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        s_rd = torch.empty(n + 16, dtype=torch.int32, device=dev)
        synth.generate_device(lib, sp, s_fa.data_ptr(), s_rd.data_ptr())
        torch.cuda.synchronize()
        sparams = api.make_params(**synth.config_flags(cfg))
        one = [(s_rd.data_ptr(), s_fa.data_ptr(), n)]
        for _ in range(2):
            pool.run(sparams, one)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            r1 = pool.run(sparams, one)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / reps
        out["single_chromosome_configs"][label] = {"ms": round(dt1 * 1e3, 3), "bases_per_s": round(n / dt1, 1), "calls": len(r1[0].calls("calls"))}
        # the same chromosome from pinned host memory through rsi_hot_run: H2D of depth + FASTA (5 B/base) inside the time
        h_rd = torch.empty(n, dtype=torch.int32).pin_memory()
        h_fa = torch.empty(n, dtype=torch.uint8).pin_memory()
        h_rd.copy_(s_rd[:n]); h_fa.copy_(s_fa[:n])
        torch.cuda.synchronize()
        del s_fa, s_rd
        depth_np, fasta_np = h_rd.numpy(), h_fa.numpy()
        hot.run(sparams, depth_np, fasta_np)
        t2 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            r2 = hot.run(sparams, depth_np, fasta_np)
        dt2 = (time.perf_counter() - t2) / reps
        out["t_device_h2d"][label] = {"ms": round(dt2 * 1e3, 3), "bases_per_s": round(n / dt2, 1), "calls": len(r2.calls("calls")),
                                      "note": "rsi_hot_run: pinned host depth + FASTA -> results on the host, one context"}
        if cfg == 2:
            host_case = (fasta_np.copy(), depth_np.copy())
        else:
            host_case250 = (fasta_np.copy(), depth_np.copy())
        del h_rd, h_fa
    hot.close()
    # ---- the genome from pinned host memory through the pool (rsi_pool_run_host): every worker moves its chromosome over its
    # own stream, transfers of some chromosomes beside the kernels of others; PCIe-bound at 5 B/base ----
    try:
        out["t_device_h2d_genome"] = genome_from_host(lib, pool, dev, args)
    except Exception as e:
        out["t_device_h2d_genome"] = {"error": str(e)[:200]}
    # ---- the envelope the byte path does not cover, on one 60 Mb chromosome each: -NOGC (no GC pass: 13.7
    # algorithmic B/base), wide bins (m = 201: the int32 K4), 300x coverage (int32 K3 / K4 with windows that follow the depth) ----
    try:
        out["fallback_envelope"] = fallback_envelope(lib, pool, dev)
    except Exception as e:
        out["fallback_envelope"] = {"error": str(e)[:200]}
    # ---- configs[4] (3 Gb at 60x, -m 51 -MED -cap 4) as a side pass: a driver-timed number for the other 3 Gb
    # configuration, with its scan kernel's rate (Lmax 196 there) ----
    try:
        out["configs[4]_side_pass"] = config5_side_pass(lib, pool, dev, args)
    except Exception as e:   # a side measurement must not take the bench line down
        out["configs[4]_side_pass"] = {"error": str(e)[:200]}
    pool.set_timing(3)
    # ---- t_e2e: the command line on the 60 Mb chromosome as files (FASTA + .fai, 700 MB of "pos depth" text) ----
    try:
        fasta_np, depth_np = host_case
        exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            fa, rdf = synth.write_case_files(lib, fasta_np, depth_np, d)
            best, calls = None, 0
            for _ in range(2):
                t3 = time.perf_counter()
                r = subprocess.run([exe, "rsi", "-f", fa, "-d", rdf, "-c", "chrS", "-o", os.path.join(d, "out.txt"), "-np"], capture_output=True, timeout=300)
                dt3 = time.perf_counter() - t3
                if r.returncode != 0:
                    raise RuntimeError(r.stderr.decode()[-300:])
                best = dt3 if best is None else min(best, dt3)
                calls = sum(1 for l in open(os.path.join(d, "out.txt")) if not l.startswith("#"))
            out["t_e2e"] = {"s": round(best, 3), "bases_per_s": round(depth_np.size / best, 1), "calls": calls,
                            "text_bytes": os.path.getsize(rdf),
                            "note": "rsicnv rsi -f REF -d RDFILE -c chrS on configs[1] (process start, FASTA, depth text parse, device path, output file)"}
    except Exception as e:   # a side measurement must not take the bench line down
        out["t_e2e"] = {"error": str(e)[:200]}
    # ---- the same for configs[2]: 250 Mb, 2.9 GB of depth text (loaddata.cpp:473-539, rsi.cpp:2069-2259) ----
    try:
        import shutil
        fasta_np, depth_np = host_case250
        if shutil.disk_usage("/tmp").free < 6 * (1 << 30):
            raise RuntimeError("less than 6 GB free under /tmp")
        exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            tw = time.perf_counter()
            fa, rdf = synth.write_case_files(lib, fasta_np, depth_np, d)
            tw = time.perf_counter() - tw
            best, calls = None, 0
            for _ in range(2):
                t3 = time.perf_counter()
                r = subprocess.run([exe, "rsi", "-f", fa, "-d", rdf, "-c", "chrS", "-o", os.path.join(d, "out.txt"), "-np"], capture_output=True, timeout=600)
                dt3 = time.perf_counter() - t3
                if r.returncode != 0:
                    raise RuntimeError(r.stderr.decode()[-300:])
                best = dt3 if best is None else min(best, dt3)
                calls = sum(1 for l in open(os.path.join(d, "out.txt")) if not l.startswith("#"))
            out["t_e2e_250Mb"] = {"s": round(best, 3), "bases_per_s": round(depth_np.size / best, 1), "calls": calls, "text_bytes": os.path.getsize(rdf),
                                  "files_written_in_s": round(tw, 1),
                                  "note": "rsicnv rsi -f REF -d RDFILE -c chrS on configs[2] (process start, FASTA, depth text parse from the page cache, device path, output file)"}
    except Exception as e:
        out["t_e2e_250Mb"] = {"error": str(e)[:200]}
    return out


def host_memory_budget():
    """Bytes of host memory this process may still take (cgroup limit and MemAvailable), or 0 when unknown."""
    avail = 0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        return 0
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
        if lim != "max":
            avail = min(avail, int(lim) - cur)
    except (OSError, ValueError):
        pass
    return max(avail, 0)


def genome_from_host(lib, pool, dev, args):
    """configs[3]'s chromosomes in pinned host memory -> results on the host: as much of the genome as a third of the free host
    memory holds (5 bytes per base pinned), longest chromosomes first."""
    import torch
    from rsicnv_amd import api, synth
    budget = host_memory_budget() // 3
    plans = sorted((synth.config_plan(4, chrom=c) for c in range(24)), key=lambda p: -p["n"])
    params = api.make_params(**synth.config_flags(4))
    host, total = [], 0
    for p in plans:
        need = 5 * p["n"] + 4096
        if budget < need:
            break
        budget -= need
        d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device=dev)
        d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device=dev)
        synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
        h_rd = torch.empty(p["n"], dtype=torch.int32, pin_memory=True)
        h_fa = torch.empty(p["n"], dtype=torch.uint8, pin_memory=True)
        h_rd.copy_(d_rd[:p["n"]]); h_fa.copy_(d_fa[:p["n"]])
        torch.cuda.synchronize()
        del d_fa, d_rd
        host.append((h_rd, h_fa, p["n"]))
        total += p["n"]
    if not host:
        return {"skipped": "not enough free host memory for one chromosome"}
    chroms = [(a.data_ptr(), b.data_ptr(), n) for a, b, n in host]
    pool.set_timing(0)
    pool.run(params, chroms, host=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        res = pool.run(params, chroms, host=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"chromosomes": len(host), "bases": total, "ms": round(dt * 1e3, 2), "bases_per_s": round(total / dt, 1),
            "input_GBps": round(5.0 * total / dt / 1e9, 1), "link_bytes_per_base": 2.0, "link_GBps": round(2.0 * total / dt / 1e9, 1),
            "calls": sum(len(r.calls("calls")) for r in res),
            "note": "rsi_pool_run_host: pinned host depth (int32) + FASTA of the 3 Gb genome's chromosomes (as many as a third of the free host "
                    "memory holds, longest first) -> results on the host; every worker narrows its chromosome's depth to bytes on the host (AVX2, "
                    "the long ones on two threads), sends 1 + 1 bytes per base over PCIe on its own stream and widens on the device; "
                    "input_GBps = the 5 bytes per base of caller memory consumed per second"}


def fallback_envelope(lib, pool, dev):
    """One 60 Mb chromosome alone through the pool under flags / coverage that leave the byte kernels: ms, bases/s, and the
    per-base kernels' HIP-event times (one extra pass)."""
    import torch
    from rsicnv_amd import api, synth
    out = {}
    cases = (("30x default (-m 101 -NB): the byte path, for comparison", dict(mean=30.0), dict()),
             ("-NOGC", dict(mean=30.0), dict(gcadjust=0)),
             ("-m 201", dict(mean=30.0), dict(m=201)),
             ("300x coverage", dict(mean=300.0), dict()))
    for label, plan_kw, flag_kw in cases:
        p = synth.make_plan(60_000_000, 0x5EED0E00, model=1, n_events=9, gaps=1, **plan_kw)
        d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device=dev)
        d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device=dev)
        synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
        torch.cuda.synchronize()
        params = api.make_params(**flag_kw)
        one = [(d_rd.data_ptr(), d_fa.data_ptr(), p["n"])]
        pool.set_timing(0)
        for _ in range(2):
            pool.run(params, one)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            r = pool.run(params, one)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        saved = pool.times
        pool.reset_times()
        pool.set_timing(2)
        pool.run(params, one, collect_times=True)
        torch.cuda.synchronize()
        kt = {k: round(v[0] * 1e3) for k, v in pool.kernel_table().items()}
        pool.times = saved
        pool.set_timing(0)
        out[label] = {"ms": round(dt * 1e3, 3), "bases_per_s": round(p["n"] / dt, 1), "calls": len(r[0].calls("calls")), "per_base_kernel_us": kt}
        del d_fa, d_rd
    return out


def config5_side_pass(lib, pool, dev, args, steps=10, warmup=5):
    """BASELINE.json configs[4] through the same pool: ms per genome, whole-path fraction (24.3 B/base at m = 51), the scan
    kernel's window evaluations per second.  Never part of `value`."""
    import torch
    from rsicnv_amd import api, synth
    flags = synth.config_flags(5)
    params = api.make_params(**flags)
    data, total = [], 0
    for c in range(24):
        p = synth.config_plan(5, chrom=c)
        d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device=dev)
        d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device=dev)
        synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
        data.append((d_rd, d_fa, p["n"]))
        total += p["n"]
    torch.cuda.synchronize()
    chrom_args = [(a.data_ptr(), b.data_ptr(), n) for a, b, n in data]
    pool.set_timing(0)
    for _ in range(warmup):
        pool.run(params, chrom_args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending, all_res = [], []
    for _ in range(steps):   # as the main line: at most --inflight genomes queued in the pool
        pending.append(pool.submit(params, chrom_args))
        if len(pending) >= max(1, args.inflight):
            all_res.append(pool.wait(pending.pop(0)))
    while pending:
        all_res.append(pool.wait(pending.pop(0)))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = all_res[-1]
    # every timed step's results, hashed as the main line hashes a step (rows in chromosome order, then chromosome median / SD)
    # and compared with the hash of the reference's rows (tests/golden/genome_rows.json); the results are on the host when the
    # timer stops, the text is made afterwards
    import hashlib
    hashes = []
    for rs in all_res:
        h = hashlib.sha256()
        for c, r in enumerate(rs):
            for row in r.format_rows(f"chr{c + 1}"):
                h.update(row.encode()); h.update(b"\n")
        for c, r in enumerate(rs):
            h.update(repr((c, float(r.stats["RDmedian"]), float(r.stats["RDsd"]))).encode())
        hashes.append(h.hexdigest())
    rows_ref = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "genome_rows.json")) as f:
            rows_ref = json.load(f).get("config5")
    except OSError:
        pass
    rows_match = None if rows_ref is None else all(h == rows_ref["rows_sha256"] for h in hashes)
    del all_res
    ncalls = sum(len(r.calls("calls")) for r in res)
    evals = sum(4 * int(r.stats["Lmax"]) * int(r.stats["nbins"]) for r in res)
    whole = WHOLE_PATH_BYTES_PER_BASE[51] * total / dt / 1e9
    saved = pool.times
    pool.reset_times()
    pool.set_timing(1)
    pool.run(params, chrom_args, collect_times=True)
    torch.cuda.synchronize()
    table = pool.kernel_table()
    pool.times = saved
    pool.set_timing(0)
    out = {"workload": "24 synthetic chromosomes totalling 3 Gb, 60x, -m 51 -MED -cap 4 (configs[4])", "flags": flag_string(flags),
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3), "bases_per_s": round(total / dt, 1), "calls_per_genome": ncalls,
           "steps_identical": len(set(hashes)) == 1, "rows_sha256": hashes[0], "rows_match_reference": rows_match,
           "whole_path": {"algorithmic_bytes_per_base": WHOLE_PATH_BYTES_PER_BASE[51], "achieved": round(whole, 1), "frac": round(whole / HBM_PEAK_GBS, 4)},
           "kernel_ms_all_launches_extra_pass": {k: round(v[0], 3) for k, v in sorted(table.items(), key=lambda kv: -kv[1][0])[:10]}}
    if "rsi_scan" in table and table["rsi_scan"][0] > 0:
        out["scan"] = {"window_evaluations_per_step": int(evals), "kernel_ms_per_step_in_situ": round(table["rsi_scan"][0], 3),
                       "evaluations_per_s": round(evals / (table["rsi_scan"][0] * 1e-3), 1)}
    del data
    return out


def flag_string(f):
    return f"-m {f['m']} " + ("-NB" if f["trans"] == 0 else "-MED" if f["trans"] == 1 else "-ALL") + f" -cap {f['cap']:g}" + \
        ("" if f["gcadjust"] else " -NOGC")


def workload_name(args):
    names = {2: "synthetic 60 Mb chromosome, 30x Poisson depth (configs[1])",
             3: "synthetic 250 Mb chromosome, 30x gamma-Poisson depth with GC dependence (configs[2])",
             4: "24 synthetic chromosomes totalling 3 Gb, 30x (configs[3]; north_star's 3 Gb genome)",
             5: "24 synthetic chromosomes totalling 3 Gb, 60x, -m 51 -MED -cap 4 (configs[4])"}
    s = names[args.config]
    if args.scale != 1.0:
        s += f" [scaled x{args.scale}: NOT the metric's configuration]"
    return s


def host_cpu_share():
    """CPUs this process may really use: the cgroup quota when there is one (a GPU box gives a 1-GPU job 16 of its 256
    CPUs), else the affinity mask; at most 32 (the all-cores sample stays bounded)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


_CPU_CHILD = r"""
import sys, time, numpy as np
sys.path.insert(0, sys.argv[1])
import oracle
d = np.load(sys.argv[2])
p = oracle.make_params(**eval(sys.argv[3]))
t0 = time.perf_counter()
if oracle.ref_available():
    nc, _ = oracle.Ref().run_timed(p, d["depth"], d["fasta"])
else:
    nc = oracle.Oracle().run(p, d["depth"], d["fasta"], snapshots=False)
print(time.perf_counter() - t0, nc)
"""


def cpu_baseline(lib, args, flags):
    """Reference (or the oracle port) on one chromosome of the same model, compute-only, one core; then the same chromosome
    in one process per host core at once (the reference has no threads: processes are how it uses a machine)."""
    import numpy as np
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
            subprocess.run(["make", "-f", "oracle/Makefile", "oracle/librsi_oracle.so"], cwd=ROOT, check=True)
        O = oracle.Oracle()
        nc = O.run(p, depth, fasta, snapshots=False)
        stages = list(O.f64("stage_s"))
        kind = "port"
    dt = time.perf_counter() - t0
    out = {"value": round(n / dt, 1), "unit": "bases/s", "cores": 1, "kind": kind,
           "sample": f"one {n/1e6:.0f} Mb chromosome of the same depth model and flags, compute-only "
                     f"(arrays in memory -> calls), {dt:.1f} s, {nc} calls",
           "stage_s": [round(s, 3) for s in stages]}
    # ---- all host cores: one process per core, each the same chromosome (bounded: one chromosome per core) ----
    try:
        cores = host_cpu_share()
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            path = os.path.join(d, "case.npz")
            np.savez(path, depth=depth, fasta=fasta)
            t1 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, "-c", _CPU_CHILD, ROOT, path, repr(flags)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                     for _ in range(cores)]
            outs = [pr.communicate(timeout=600) for pr in procs]
            wall = time.perf_counter() - t1
            if any(pr.returncode != 0 for pr in procs):
                raise RuntimeError(outs[0][1].decode()[-200:])
            each = [float(o[0].split()[0]) for o in outs]
        out["all_cores"] = {"value": round(cores * n / wall, 1), "unit": "bases/s", "cores": cores, "kind": kind,
                            "sample": f"{cores} processes at once, one {n/1e6:.0f} Mb chromosome each: {wall:.1f} s wall "
                                      f"(compute {min(each):.1f}-{max(each):.1f} s per process, process start and input load included in the wall time)"}
    except Exception as e:
        out["all_cores"] = {"error": str(e)[:200]}
    return out


if __name__ == "__main__":
    main()
