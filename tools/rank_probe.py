"""What one rank of a sharded-genome run does, measured on ONE GPU: for world sizes 2 / 4 / 8 every rank's share of the
24-chromosome genome (rsicnv_amd.dist.lpt_assign) goes through the same pool (20 workers, bench.py's default on a 16-core box), one share after the other, queued
the way bench.py queues its steps (as many genomes in flight as keep the workers busy: 2 for a whole genome, up to 6 for a
rank's share); the slowest share is the step time an N-GPU run would see (without the all_gather, ~0.1 ms).  --serial: one
run at a time (the latency of a share, what bench.py measured before round 3).  Usage:
  python tools/rank_probe.py [--worlds 2 4 8] [--rounds 5] [--env KEY=VALUE ...]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--workers", type=int, default=20)
    ap.add_argument("--serial", action="store_true")
    args = ap.parse_args()
    for kv in args.env:
        k, _, v = kv.partition("=")
        os.environ[k] = v
    import torch
    from rsicnv_amd import api, synth, dist as rd
    lib = api.load_library()
    torch.cuda.set_device(0)
    params = api.make_params(**synth.config_flags(args.config))
    data, lengths = [], []
    for c in range(24):
        p = synth.config_plan(args.config, chrom=c)
        d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda")
        d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
        synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
        data.append((d_rd, d_fa, p["n"]))
        lengths.append(p["n"])
    torch.cuda.synchronize()
    pool = api.RsiPool(0, args.workers)
    pool.set_timing(0)
    allc = [(a.data_ptr(), b.data_ptr(), n) for a, b, n in data]

    def per_genome_ms(chroms, rounds):
        """rounds genomes of `chroms`, queued as bench.py queues its steps; ms per genome"""
        infl = 1 if args.serial else max(2, min(6, (args.workers + len(chroms) - 1) // len(chroms)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = []
        for _ in range(rounds):
            pending.append(pool.submit(params, chroms))
            if len(pending) >= infl:
                pool.wait(pending.pop(0))
        while pending:
            pool.wait(pending.pop(0))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / rounds * 1e3, infl

    def best_of(chroms, loops=3):
        """the fastest of `loops` timed loops (a loop now and then is 10-25 % slow: 12.8 - 16.0 ms for the whole genome on one box)"""
        per_genome_ms(chroms, 3)
        runs = [per_genome_ms(chroms, 4 * args.rounds) for _ in range(loops)]
        return min(r[0] for r in runs), runs[0][1], [round(r[0], 2) for r in runs]

    whole, infl, loops = best_of(allc)
    print(f"world 1: {whole:.2f} ms per genome ({infl} in flight; loops {loops})", flush=True)
    for world in args.worlds:
        parts = rd.lpt_assign(lengths, world)
        worst, per = 0.0, []
        for part in parts:
            mine = [allc[i] for i in part]
            ms, infl, _ = best_of(mine)
            per.append((len(part), sum(lengths[i] for i in part) / 1e6, ms))
            worst = max(worst, ms)
        print(f"world {world}: slowest share {worst:.2f} ms per genome ({infl} in flight) -> x{whole / worst:.2f} of one GPU; shares (chromosomes, Mb, ms): "
              + " ".join(f"({k},{mb:.0f},{ms:.2f})" for k, mb, ms in per), flush=True)


if __name__ == "__main__":
    main()
