"""In-process A/B of pipeline variants on one GPU box (box-to-box noise is ~10 %, so variants
are only comparable inside one process).  Usage:
  python tools/ab_bench.py --variants "w12:workers=12" "w24:workers=24" "host:workers=12,RSI_HOT_HOST_CANDIDATES=1" --rounds 4
Each variant: `workers=` picks the pool, the other KEY=VALUE pairs are exported before the step
(the library reads its RSI_HOT_* switches at run time).  Prints mean / min ms per variant."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="+", required=True)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--phases", action="store_true")
    args = ap.parse_args()
    import torch
    from rsicnv_amd import api, synth
    lib = api.load_library()
    torch.cuda.set_device(0)
    params = api.make_params(**synth.config_flags(args.config))
    chroms = [0] if args.config in (2, 3) else list(range(24))
    data = []
    for c in chroms:
        p = synth.config_plan(args.config, chrom=c)
        d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda")
        d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
        synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
        data.append((d_rd, d_fa, p["n"]))
    torch.cuda.synchronize()
    chrom_args = [(a.data_ptr(), b.data_ptr(), n) for a, b, n in data]
    variants = []
    pools = {}
    for v in args.variants:
        name, _, spec = v.partition(":")
        kv = dict(x.split("=", 1) for x in spec.split(",") if x)
        w = int(kv.pop("workers", 12))
        iso = kv.pop("RSI_HOT_ISOLATE_STREAMING", "0")
        ms = kv.pop("RSI_HOT_STREAMERS", "")
        key = (w, iso, ms)
        if key not in pools:
            os.environ["RSI_HOT_ISOLATE_STREAMING"] = iso
            if ms:
                os.environ["RSI_HOT_STREAMERS"] = ms
            else:
                os.environ.pop("RSI_HOT_STREAMERS", None)
            pools[key] = api.RsiPool(0, w)
            pools[key].set_timing(True)
        variants.append((name, pools[key], kv))
    keys = sorted({k for _, _, kv in variants for k in kv if k not in ("timing", "rows")})
    times = {name: [] for name, _, _ in variants}
    tables = {name: api.RsiBatchTimes() for name, _, _ in variants}
    calls = {}

    def throttled():
        try:
            d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
            return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0))
        except Exception:
            return 0, 0, 0

    notes = {name: [] for name, _, _ in variants}

    def run(name, pool, kv, timed):
        for k in keys:   # a switch a variant does not name is "0" for it; the value "-" removes it from the environment
            v = kv.get(k, "0")
            if v == "-":
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        pool.set_timing(int(kv.get("timing", "1")))   # 1: HIP events around every launch; 3: around the dominant kernel only (bench.py's timed steps)
        timed = timed and kv.get("timing", "1") != "0"
        pool.times = tables[name]
        torch.cuda.synchronize()
        th0 = throttled()
        t0 = time.perf_counter()
        res = pool.run(params, chrom_args, collect_times=timed)
        if kv.get("rows", "0") != "0":   # rank 0's share of a bench.py step: blocks -> ordered rows
            from rsicnv_amd import dist as rd
            ids = list(range(len(res)))
            merged = rd.unpack_blocks([rd.pack_results(ids, res, len(res))])
            rd.format_rows(lib, merged, [f"chr{c + 1}" for c in ids])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        th1 = throttled()
        if timed:
            notes[name].append(f"{dt:.0f}ms:thr{th1[0]-th0[0]}/{(th1[1]-th0[1])/1e3:.0f}ms/cpu{(th1[2]-th0[2])/1e3/dt:.1f}")
        sig = tuple((len(r.calls("calls")), r.stats["RDmedian"]) for r in res)
        full = [[(c["start"], c["end"], c["type"]) for c in r.calls("calls")] for r in res]
        calls.setdefault("ref", full)
        if full != calls["ref"]:
            print(f"!! variant {name} produced different calls", flush=True)
        return dt

    for name, pool, kv in variants:   # warm-up
        run(name, pool, kv, False)
        run(name, pool, kv, False)
    for r in range(args.rounds):
        for name, pool, kv in variants:
            times[name].append(run(name, pool, kv, True))
    for name, pool, kv in variants:
        t = times[name]
        print(f"{name:>16s}: mean {sum(t)/len(t):7.2f} ms  min {min(t):7.2f}  max {max(t):7.2f}   ({' '.join(f'{x:.0f}' for x in t)})", flush=True)
    for name, _, _ in variants:
        print(f"    {name} steps (wall : throttled periods / throttled thread-ms / mean busy CPUs): " + " ".join(notes[name]), flush=True)
    if args.phases:
        for name, pool, kv in variants:
            n = len(times[name])
            pool.times = tables[name]
            ph = sorted(pool.phase_table().items(), key=lambda x: -x[1])
            kt = sorted(pool.kernel_table().items(), key=lambda x: -x[1][0])
            print(f"--- {name}: phases ms/step: " + ", ".join(f"{k}={v/n:.1f}" for k, v in ph[:24]))
            print(f"--- {name}: kernels ms/step: " + ", ".join(f"{k}={v[0]/n:.1f}" for k, v in kt[:16]))


if __name__ == "__main__":
    main()
