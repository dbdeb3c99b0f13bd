"""Soak of the pool's run queue on the GPU: the 3 Gb genome (or a scaled one) submitted back to back for --seconds, --inflight
genomes queued at a time, from two submitting threads; every genome's rows must hash like the first one's.
usage: python tools/soak_queue.py [--seconds 240] [--inflight 3] [--scale 1.0]"""
import argparse, hashlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--workers", type=int, default=16)
    args = ap.parse_args()
    import torch
    from rsicnv_amd import api, synth
    lib = api.load_library()
    torch.cuda.set_device(0)
    params = api.make_params(**synth.config_flags(4))
    data = []
    for c in range(24):
        p = synth.config_plan(4, chrom=c, scale=args.scale)
        d_fa = torch.empty(p["n"] + 64, dtype=torch.uint8, device="cuda")
        d_rd = torch.empty(p["n"] + 16, dtype=torch.int32, device="cuda")
        synth.generate_device(lib, p, d_fa.data_ptr(), d_rd.data_ptr())
        data.append((d_rd, d_fa, p["n"]))
    torch.cuda.synchronize()
    chroms = [(a.data_ptr(), b.data_ptr(), n) for a, b, n in data]
    pool = api.RsiPool(0, args.workers)

    def digest(results):
        h = hashlib.sha256()
        for r in results:
            h.update(repr((r.stats["RDmedian"], r.stats["RDsd"], [(c["start"], c["end"], c["type"], c["score"]) for c in r.calls("calls")])).encode())
        return h.hexdigest()

    first = digest(pool.run(params, chroms))
    t_end = time.time() + args.seconds
    done, bad = [0, 0], []
    lock = threading.Lock()

    def client(k):
        pending = []
        while time.time() < t_end and not bad:
            pending.append(pool.submit(params, chroms if k == 0 else chroms[::-1]))
            if len(pending) >= args.inflight:
                res = pool.wait(pending.pop(0))
                d = digest(res if k == 0 else res[::-1])
                with lock:
                    done[k] += 1
                    if d != first:
                        bad.append((k, done[k]))
        for h in pending:
            res = pool.wait(h)
            if digest(res if k == 0 else res[::-1]) != first:
                bad.append((k, -1))
            done[k] += 1

    th = [threading.Thread(target=client, args=(k,)) for k in range(2)]
    t0 = time.time()
    for t in th: t.start()
    last = 0
    while any(t.is_alive() for t in th):
        time.sleep(20)
        print(f"[{time.time() - t0:5.0f} s] genomes done {sum(done)} (+{sum(done) - last}), mismatches {len(bad)}", flush=True)
        last = sum(done)
    for t in th: t.join()
    dt = time.time() - t0
    print(f"soak: {sum(done)} genomes in {dt:.0f} s ({dt / max(1, sum(done)) * 1e3:.2f} ms per genome), mismatches: {bad}", flush=True)
    pool.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
