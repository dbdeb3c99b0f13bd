"""What one rank of a sharded-genome run does, measured on ONE GPU: for world sizes 2 / 4 / 8 every rank's share of the
24-chromosome genome (rsicnv_amd.dist.lpt_assign) goes through the same 12-worker pool, one share after the other; the
slowest share is the step time an N-GPU run would see (without the all_gather, ~0.1 ms).  Usage:
  python tools/rank_probe.py [--worlds 2 4 8] [--rounds 5] [--env KEY=VALUE ...]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--env", nargs="*", default=[])
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
    pool = api.RsiPool(0, 12)
    pool.set_timing(0)
    allc = [(a.data_ptr(), b.data_ptr(), n) for a, b, n in data]
    for _ in range(2):
        pool.run(params, allc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.rounds):
        pool.run(params, allc)
    torch.cuda.synchronize()
    whole = (time.perf_counter() - t0) / args.rounds * 1e3
    print(f"world 1: {whole:.2f} ms per genome", flush=True)
    for world in args.worlds:
        parts = rd.lpt_assign(lengths, world)
        worst, per = 0.0, []
        for part in parts:
            mine = [allc[i] for i in part]
            pool.run(params, mine)
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.rounds):
                t0 = time.perf_counter()
                pool.run(params, mine)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            med = ts[len(ts) // 2]
            per.append((len(part), sum(lengths[i] for i in part) / 1e6, med))
            worst = max(worst, med)
        print(f"world {world}: slowest share {worst:.2f} ms -> x{whole / worst:.2f} of one GPU; shares (chromosomes, Mb, ms): "
              + " ".join(f"({k},{mb:.0f},{ms:.2f})" for k, mb, ms in per), flush=True)


if __name__ == "__main__":
    main()
