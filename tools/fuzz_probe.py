"""Random chromosomes x random flags through the library and the oracle (the oracle in a child process: where the reference would exit
or abort, the restatement may too).  Every array, scalar and list is compared.  usage: fuzz_probe.py [cases] [seed]"""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import oracle
d = np.load(sys.argv[2])
fl = json.loads(sys.argv[3])
O = oracle.Oracle()
rc = O.run(oracle.make_params(**fl), d["depth"], d["fasta"])
out = {"rc": rc}
if rc >= 0:
    pre = "med" if fl.get("trans", 0) == 1 else "nb"
    arrs = {k: O.i32(k) for k in ("noncode", "rd_concat", "binmedint", pre + "_status1", pre + "_status1f", pre + "_status2")}
    if fl.get("gcadjust", 1): arrs["rd_gc"] = O.i32("rd_gc")
    arrs["binnb"] = O.f32("binnb")
    np.savez(sys.argv[4], **arrs)
    keys = ("start", "end", "type", "geno", "status", "length", "qscore", "score", "p1", "cnvmed", "cnvsd", "cnviqr", "refmed", "refsd", "refiqr")
    for which in ("blocks", "calls_raw", "calls"):
        out[which] = [[c[k] for k in keys] for c in O.calls(which)]
    out["chrom"] = list(O.f64("chrom")); out["nb"] = list(O.f64("nb")); out["scan"] = list(O.f64("scan_" + pre))
print("RESULT " + json.dumps(out))
"""
KEYS = ("start", "end", "type", "geno", "status", "length", "qscore", "score", "p1", "cnvmed", "cnvsd", "cnviqr", "refmed", "refsd", "refiqr")


def run(ncases, seed, max_bins=250_000, deep=False):
    """Returns the number of cases that differ (or where only one side refused).  deep: coverage of 120x / 300x under a cap, every
    bin-size class of the 16-bit compaction kernel (K4w: 2, 4, 8 and 16 threads per bin)."""
    from conftest import make_case
    from rsicnv_amd import api
    rng = np.random.default_rng(seed)
    lib = api.load_library()
    hot = api.RsiHot(0)
    bad = 0
    for case in range(ncases):
        n = int(rng.choice([60_000, 150_000, 400_000, 900_000, 2_000_000])) + int(rng.integers(0, 40))
        model = int(rng.integers(0, 2))
        mean = float(rng.choice([8, 15, 30, 30, 60, 120, 300]))
        if deep: mean = float(rng.choice([120, 300, 300]))
        crowded = rng.random() < 0.3
        plan_kw = dict(n=n, seed=int(rng.integers(1, 1 << 30)), model=model, mean=mean, n_events=int(rng.integers(20, 60)) if crowded else int(rng.integers(1, 12)),
                       gaps=int(rng.integers(0, 4)), min_len=600 if crowded else 3000, max_len=int(rng.choice([3000, 8000])) if crowded else int(rng.choice([5000, 20000, 60000])),
                       end_n=int(rng.choice([0, 3000, 10000])), gap_len=int(rng.choice([200, 3000, 9000])))
        fl = dict(m=int(rng.choice([11, 51, 101, 101, 201, 439])), trans=int(rng.choice([0, 0, 1, 2])), cap=float(rng.choice([-1.0, 2.0, 4.0, 4.0])),
                  gcadjust=int(rng.choice([0, 1, 1, 1])), merge=int(rng.choice([0, 1, 1])))
        if deep: fl.update(m=int(rng.choice([11, 51, 101, 201, 439])), cap=float(rng.choice([2.0, 4.0, 8.0])))
        if rng.random() < 0.5:   # the rarely touched knobs (rsi.cpp:34-98): score factor, MED threshold, neighbourhood size, minimum length,
                                 # margin, thinning budget, p-value bar
            fl.update(epsilon=float(rng.choice([0.5, 1.5, 3.0])), chklen=float(rng.choice([1.5, 2.5, 4.0])), minmlen=float(rng.choice([2.01, 3.01, 6.0])),
                      buffer=float(rng.choice([0.0, 0.05, 0.2])), maxchkbp=int(rng.choice([2000, 20000, 100000])), p=float(rng.choice([0.01, 0.05, 0.2])))
            if fl["trans"] == 1 and rng.random() < 0.5:
                fl["threshold"] = float(rng.choice([0.3, 0.6]))
        if n // fl["m"] < 1200:
            fl["m"] = 11 if n < 100_000 else 51
        if n // fl["m"] > max_bins:      # the oracle's scan at -m 11 takes half a minute per 200 000 bins
            fl["m"] = 101
        try:
            _, fasta, depth = make_case(lib, plan_kw)
        except Exception as e:
            print(f"case {case}: generator refused {plan_kw}: {e}"); continue
        extra = int(rng.integers(0, 4))
        depth = depth.copy(); fasta = fasta.copy()
        if extra == 1:   # a pile-up: a short stretch at 50x the depth
            a = int(rng.integers(n // 10, n - n // 10)); depth[a:a + 300] *= 50
        if extra == 2:   # uncovered stretches
            for a in rng.integers(0, n - 5000, size=5): depth[a:a + int(rng.integers(50, 4000))] = 0
        if extra == 3:   # assembly gaps
            for a in rng.integers(5000, n - 5000, size=int(rng.integers(20, 300))):
                ln = int(rng.integers(1, 300)); fasta[a:a + ln] = ord("N"); depth[a:a + ln] = 0
        tag = f"case {case}: n={n} model={model} mean={mean:g} events={plan_kw['n_events']} extra={extra} flags={fl}"
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            path, outp = os.path.join(td, "c.npz"), os.path.join(td, "o.npz")
            np.savez(path, depth=depth, fasta=fasta)
            t0 = time.time()
            try:
                r = subprocess.run([sys.executable, "-c", CHILD, ROOT, path, json.dumps(fl), outp], capture_output=True, text=True, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
                orc = json.loads(line[0][7:]) if line else {"rc": f"died ({r.returncode})"}
            except subprocess.TimeoutExpired:
                orc = {"rc": "timeout"}
            t_or = time.time() - t0
            try:
                res = hot.run(api.make_params(**fl), depth, fasta)
                lib_err = None
            except api.RsiError as e:
                res, lib_err = None, str(e)[:100]
            orc_ok = isinstance(orc.get("rc"), int) and orc["rc"] >= 0
            if res is None or not orc_ok:
                both = res is None and not orc_ok
                print(f"{tag}: {'both refuse' if both else 'ONE-SIDED'} (lib: {lib_err or 'ok'}; oracle: {orc['rc']})", flush=True)
                bad += 0 if both else 1
                continue
            og = np.load(outp)
            pre = "med" if fl["trans"] == 1 else "nb"
            diffs = []
            if not np.array_equal(res.noncode, og["noncode"]): diffs.append("noncode")
            st_low = res.stats["RDmedian"] < 5
            for name, key in (("rd_concat", "rd_concat"), ("binmedint", "binmedint"), ("status1", pre + "_status1"), ("status1f", pre + "_status1f"), ("status2", pre + "_status2")):
                try:
                    a = hot.fetch(name)
                except (KeyError, api.RsiError):
                    a = np.zeros(0, dtype=np.int32)
                if name != "rd_concat" and og[key].size == 0 and st_low: continue   # "Read depths too low" (rsi.cpp:1809-1813): the reference returns before it bins anything
                if a.shape != og[key].shape or not np.array_equal(a, og[key]): diffs.append(f"{name} {a.shape} vs {og[key].shape}")
            if fl["gcadjust"] and not np.array_equal(hot.fetch("rd_gc"), og["rd_gc"]): diffs.append("rd_gc")
            try:
                nbv = hot.fetch("binnb")
            except KeyError:      # a median depth below 5: the reference returns before any transform (rsi.cpp:1809-1813)
                nbv = np.zeros(0, dtype=np.float32)
            if nbv.shape != og["binnb"].shape or not np.allclose(nbv, og["binnb"], rtol=1e-6, atol=0): diffs.append(f"binnb {nbv.shape} vs {og['binnb'].shape}")
            st = res.stats
            if st["RDmedian"] != orc["chrom"][0] or abs(st["RDsd"] - orc["chrom"][1]) > 1e-12 * max(1, abs(orc["chrom"][1])): diffs.append(f"chrom {st['RDmedian']},{st['RDsd']} vs {orc['chrom'][:2]}")
            if st["Lmax"] != int(orc["scan"][7]): diffs.append(f"Lmax {st['Lmax']} vs {orc['scan'][7]}")
            for which in ("blocks", "calls_raw", "calls"):
                got = [[c[k] for k in KEYS] for c in res.calls(which)]
                exp = orc[which]
                if len(got) != len(exp): diffs.append(f"{which} count {len(got)} vs {len(exp)}"); continue
                for i, (x, y) in enumerate(zip(got, exp)):
                    if x[:7] != y[:7] or not np.allclose(x[7:], y[7:], rtol=1e-6, atol=1e-300):
                        diffs.append(f"{which}[{i}] {x[:3]} vs {y[:3]}"); break
            print(f"{tag}: {'same' if not diffs else 'DIFFERENT ' + '; '.join(diffs)} ({len(orc['calls'])} calls, oracle {t_or:.1f} s)", flush=True)
            bad += 1 if diffs else 0
    print(f"fuzz: {bad} of {ncases} cases need a look")
    hot.close()
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1, deep=len(sys.argv) > 3 and sys.argv[3] == "deep")
