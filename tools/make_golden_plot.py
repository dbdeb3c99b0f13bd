#!/usr/bin/env python3
"""tests/golden/plot_cases.npz: the gnuplot data (.dat) and script (.gp) files the REFERENCE's own plot_icnv
(plotcnv.cpp:246-610, inside oracle/_ref/libref.so) writes for a handful of calls on seeded per-base arrays -- a deletion, a
wide duplication (more than plot::pts positions: the subsampled walk), a call shorter than m * minmlen, calls whose
neighbourhood is clamped at either end of the chromosome.  plot_icnv deletes its files when it is done; oracle/ref_driver.cpp
(ref_plot_icnv) hard-links them first.  Runs only where /root/reference exists; the file holds data (arrays, parameters, the
reference's output text), never reference source.  The script's dialect is the one the reference picks on this machine
(gnuplot_version() = -1 without a gnuplot: the older dialect); the newer dialect cannot be produced by the reference here
without a gnuplot stand-in and stays checked against tests/test_plot_writer.py's restatement only.
  python tools/make_golden_plot.py"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [   # name, n, seed, start, end, type, depth factor inside the call
    ("del_mid", 60_000, 11, 30_000, 31_999, 0, 0.5),
    ("dup_wide", 80_000, 12, 30_000, 36_499, 1, 1.5),
    ("del_short", 50_000, 13, 40_000, 40_100, 0, 0.0),
    ("dup_left_edge", 40_000, 14, 300, 2_500, 1, 2.0),
    ("del_right_edge", 40_000, 15, 37_000, 39_500, 0, 0.5),
]
TYPES = ["DEL", "DUP"]


def main():
    import oracle
    R = oracle.Ref()
    L = R.lib
    L.ref_plot_icnv.argtypes = [C.POINTER(oracle.Params), C.POINTER(C.c_int32), C.c_int32, C.POINTER(oracle.Call), C.c_char_p, C.c_char_p, C.c_char_p,
                                C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_double)]
    p = oracle.make_params()
    out = {"names": np.array([c[0] for c in CASES])}
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        cwd = os.getcwd()
        os.chdir(d)
        os.makedirs("plots")
        try:
            for name, n, seed, start, end, typ, factor in CASES:
                rng = np.random.default_rng(seed)
                rd = rng.poisson(30, size=n).astype(np.int32)
                rd[start - 1:end] = (rd[start - 1:end] * factor).astype(np.int32)
                rd[:50] = 0
                rd[n // 2:n // 2 + 200] = 0    # an expanded N region
                c = oracle.Call()
                c.start, c.end, c.type, c.length, c.p1 = start, end, typ, end - start + 1, 1.25e-7 * (1 + seed)
                base = f"plots/rsi_chrS_{start}_{end}_{TYPES[typ]}"
                title = f"chrS:{start}~{end} {end - start + 1} {TYPES[typ]}"
                res = (C.c_double * 2)()
                rc = L.ref_plot_icnv(C.byref(p), rd.ctypes.data_as(C.POINTER(C.c_int32)), n, C.byref(c), b"chrS", title.encode(), b"ps",
                                     (base + ".dat").encode(), (base + ".gp").encode(), (base + ".ps").encode(),
                                     (base + ".dat.keep").encode(), (base + ".gp.keep").encode(), res)
                assert rc == 0, rc
                assert not os.path.exists(base + ".dat"), "the reference did not get to the end of plot_icnv"
                out[name + "_rd"] = rd
                out[name + "_call"] = np.array([start, end, typ, end - start + 1], dtype=np.int64)
                out[name + "_p1"] = np.array([c.p1])
                out[name + "_title"] = title
                out[name + "_base"] = base
                out[name + "_rdmed"] = np.array([res[0]])
                out[name + "_version"] = np.array([res[1]])
                out[name + "_dat"] = open(base + ".dat.keep").read()
                out[name + "_gp"] = open(base + ".gp.keep").read()
                print(f"{name}: dat {len(out[name + '_dat'])} bytes, gp {len(out[name + '_gp'])} bytes, RDmed {res[0]}, gnuplot_version {res[1]}")
        finally:
            os.chdir(cwd)
    out["params"] = np.array([p.m, p.minmlen, p.chklen])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "plot_cases.npz"), **out)


if __name__ == "__main__":
    main()
