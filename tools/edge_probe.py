"""Small and odd chromosomes through the library and the oracle (the oracle in a child process: where the reference would exit or
abort, the restatement may too).  Prints one line per case; anything but `same` / `both refuse` deserves a look."""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
import oracle
d = np.load(sys.argv[2])
O = oracle.Oracle()
rc = O.run(oracle.make_params(**json.loads(sys.argv[3])), d["depth"], d["fasta"])
out = {"rc": rc}
if rc >= 0:
    out["calls"] = [[c["start"], c["end"], c["type"], c["qscore"]] for c in O.calls("calls")]
    out["raw"] = [[c["start"], c["end"], c["type"]] for c in O.calls("calls_raw")]
    out["chrom"] = list(O.f64("chrom"))
print("RESULT " + json.dumps(out))
"""


def cases():
    rng = np.random.default_rng(7)
    def base(n, lam=30, events=True):
        fa = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
        d = rng.poisson(lam, size=n).astype(np.int32)
        if events and n > 60000:
            d[n // 3:n // 3 + min(8000, n // 20)] //= 2
            d[2 * n // 3:2 * n // 3 + min(8000, n // 20)] = (d[2 * n // 3:2 * n // 3 + min(8000, n // 20)] * 3) // 2
        return fa, d
    for n in (1000, 4039, 4040, 4041, 5000, 9999, 20_011, 65_537, 131_073):
        for fl in (dict(), dict(m=11), dict(gcadjust=0), dict(m=51, trans=1)):
            yield f"n={n} {fl}", base(n), fl
    fa, d = base(300_000); fa[:] = ord("N"); d[:] = 0; fa[1000:9000] = ord("A"); d[1000:9000] = 30
    yield "mostly N", (fa, d), dict()
    fa, d = base(300_000); d[50_000:150_000] = 0
    yield "a third uncovered", (fa, d), dict()
    fa, d = base(300_000, lam=3)
    yield "depth 3 (median below 5)", (fa, d), dict()
    fa, d = base(300_000, lam=7)
    yield "depth 7 (median below 10)", (fa, d), dict()
    fa, d = base(300_000); d[1234] = 70_000
    yield "one depth of 70000, cap", (fa, d), dict()
    yield "one depth of 70000, no cap", (fa, d), dict(cap=-1.0)
    fa, d = base(300_000); d[777] = -1
    yield "a negative depth", (fa, d), dict()
    fa, d = base(300_000); fa[::3] = ord("R")
    yield "IUPAC codes", (fa, d), dict()
    fa, d = base(300_000); fa = np.char.lower(fa.view("S1")).view(np.uint8).copy()
    yield "all lower case", (fa, d), dict()
    fa, d = base(400_000); d[:] = np.where(rng.random(d.size) < 0.5, 10, 50)
    yield "two depth values only", (fa, d), dict()
    fa, d = base(400_000); d[200_000:] *= 40
    yield "second half 40x deeper, no cap", (fa, d), dict(cap=-1.0)


def main():
    import tempfile
    from rsicnv_amd import api
    hot = api.RsiHot(0)
    for name, (fa, d), fl in cases():
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            path = os.path.join(td, "c.npz")
            np.savez(path, depth=d, fasta=fa)
            try:
                r = subprocess.run([sys.executable, "-c", CHILD, ROOT, path, json.dumps(fl)], capture_output=True, text=True, timeout=300)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
                orc = json.loads(line[0][7:]) if line else {"rc": f"died ({r.returncode}) {r.stderr.strip().splitlines()[-1][:80] if r.stderr.strip() else ''}"}
            except subprocess.TimeoutExpired:
                orc = {"rc": "timeout"}
        try:
            res = hot.run(api.make_params(**fl), d, fa)
            got = {"calls": [[c["start"], c["end"], c["type"], c["qscore"]] for c in res.calls("calls")], "raw": [[c["start"], c["end"], c["type"]] for c in res.calls("calls_raw")],
                   "chrom": [res.stats["RDmedian"], res.stats["RDsd"]]}
            lib = "ok"
        except api.RsiError as e:
            got, lib = None, str(e)[:90]
        if got is not None and isinstance(orc.get("rc"), int) and orc["rc"] >= 0:
            same = got["calls"] == orc["calls"] and got["raw"] == orc["raw"] and got["chrom"][0] == orc["chrom"][0] and abs(got["chrom"][1] - orc["chrom"][1]) <= 1e-12 * max(1.0, abs(orc["chrom"][1]))
            verdict = "same" if same else f"DIFFERENT lib {got['raw'][:3]} {got['chrom']} oracle {orc['raw'][:3]} {orc['chrom'][:2]}"
            print(f"{name}: {verdict} ({len(got['calls'])} calls)", flush=True)
        elif got is None and not (isinstance(orc.get("rc"), int) and orc["rc"] >= 0):
            print(f"{name}: both refuse (lib: {lib}; oracle: {orc['rc']})", flush=True)
        else:
            print(f"{name}: ONE-SIDED lib: {lib if got is None else str(len(got['calls'])) + ' calls'}; oracle: {orc['rc']}", flush=True)


if __name__ == "__main__":
    main()
