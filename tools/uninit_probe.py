"""The chromosomes of tests/test_fuzz.py's pool test, one at a time on a single context, against the oracle -- meant to be run with
RSI_HOT_POISON=1 (every allocation filled with 0xA5), where a read of memory nobody wrote changes the result instead of finding
the driver's zero pages.  One context serves all ten (as in the test: what a run leaves
behind in the context is part of what is probed); `fresh` = a new context per chromosome.  usage: uninit_probe.py [flags index 0..3] [repeats] [fresh]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import make_case
from rsicnv_amd import api
import oracle

FLAGS = [dict(), dict(m=51, trans=1), dict(gcadjust=0, cap=2.0), dict(m=201, trans=2, cap=-1.0)]
KEYS = ("start", "end", "type", "geno", "status", "length", "qscore", "score", "p1", "cnvmed", "cnvsd", "cnviqr", "refmed", "refsd", "refiqr")


def build_cases(lib, flags):
    rng = np.random.default_rng(0xF0 + len(flags))
    cases = []
    for k in range(10):
        n = int(rng.choice([40_000, 90_000, 300_000, 800_000, 1_500_000, 3_000_000])) + int(rng.integers(0, 64))
        mean = 300.0 if k == 3 else float(rng.choice([15, 30, 60]))
        _, fasta, depth = make_case(lib, dict(n=n, seed=int(rng.integers(1, 1 << 30)), model=int(rng.integers(0, 2)), mean=mean, n_events=int(rng.integers(1, 10)),
                                              gaps=int(rng.integers(0, 3)), max_len=20000, end_n=int(rng.choice([0, 3000])), gap_len=3000))
        depth = depth.copy()
        if k == 5:
            depth[n // 2:n // 2 + 300] *= 50
        cases.append((np.ascontiguousarray(fasta), np.ascontiguousarray(depth)))
    return cases


def same(a, b):   # p1 goes through pow / exp: the last digits are the libm's, not the algorithm's
    return a == b or (isinstance(a, float) and isinstance(b, float) and ((a != a and b != b) or abs(a - b) <= 1e-9 * max(abs(a), abs(b))))


def main():
    fi = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    flags = FLAGS[fi]
    lib = api.load_library()
    cases = build_cases(lib, flags)
    params = api.make_params(**flags)
    bad = 0
    fresh = len(sys.argv) > 3 and sys.argv[3] == "fresh"
    shared = None if fresh else api.RsiHot(0)
    for i, (fasta, depth) in enumerate(cases):
        O = oracle.Oracle()
        rc = O.run(oracle.make_params(**flags), depth, fasta)
        want = {w: [[c[k] for k in KEYS] for c in O.calls(w)] for w in ("blocks", "calls_raw", "calls")} if rc >= 0 else None
        for rep in range(reps):
            hot = api.RsiHot(0) if fresh else shared
            try:
                res = hot.run(params, depth, fasta)
            except api.RsiError as e:
                print(f"case {i} rep {rep}: library refused ({e}); oracle rc {rc}", flush=True)
                if fresh: hot.close()
                continue
            if want is None:
                print(f"case {i} rep {rep}: oracle refused (rc {rc}), library ran", flush=True)
                if fresh: hot.close()
                continue
            for w in ("blocks", "calls_raw", "calls"):
                got = [[c[k] for k in KEYS] for c in res.calls(w)]
                if len(got) != len(want[w]) or any(not same(x, y) for g, o in zip(got, want[w]) for x, y in zip(g, o)):
                    bad += 1
                    print(f"case {i} rep {rep} n={depth.size}: {w} DIFFERENT ({len(got)} vs {len(want[w])} entries)", flush=True)
                    for g, o in zip(got, want[w]):
                        if any(not same(x, y) for x, y in zip(g, o)):
                            print("   lib   ", g); print("   oracle", o)
                    break
            else:
                print(f"case {i} rep {rep} n={depth.size}: same ({len(want['calls'])} calls); phases: " + ",".join(p for p, _ in hot.phase_times() if "." in p and p.split(".")[0] in ("spec", "k4j", "a5")), flush=True)
            if fresh: hot.close()
    print("DIFFERENT" if bad else "ALL SAME", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
