"""The plot writer (plot_host.cpp; plotcnv.cpp:245-610): data and script files of a call, checked on the CPU against a plain
restatement of what plot_icnv puts into them (positions, depths, the running mean of the neighbourhood, the quantile
lines), and expand_data against numpy."""
import ctypes as C
import os

import numpy as np
import pytest


def _median_like(x):
    """partition_stat_tp on ints (wufunctions.cpp:364-424): min + first bucket where the cumulated count reaches n/4, n/2, 3n/4."""
    x = np.asarray(x, dtype=np.int64)
    lo, hi = int(x.min()), int(x.max())
    if hi - lo < 1:
        m = float(x.mean())
        return lo, m, hi
    h = np.bincount(x - lo, minlength=hi - lo + 2)
    c = np.cumsum(h)
    out = []
    for r in (len(x) // 4, len(x) // 2, len(x) * 3 // 4):
        i = int(np.searchsorted(c, r, side="left"))
        out.append(lo + i)
    return out[0], out[1], out[2]


def _call(lib, start, end, typ=0, p1=1e-12):
    from rsicnv_amd import api
    c = api.RsiCall()
    c.start, c.end, c.type, c.length, c.p1 = start, end, typ, end - start + 1, p1
    return c


def test_expand_puts_the_removed_regions_back(hotlib):
    rng = np.random.default_rng(5)
    n = 5000
    regions = np.array([0, 99, 1000, 1499, 4900, 4999], dtype=np.int32)
    keep = np.ones(n, dtype=bool)
    for s, e in regions.reshape(-1, 2):
        keep[s:e + 1] = False
    rdc = rng.integers(1, 90, size=int(keep.sum()), dtype=np.int32)
    out = np.full(n, -7, dtype=np.int32)
    hotlib.rsi_plot_expand.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    assert hotlib.rsi_plot_expand(rdc.ctypes.data, rdc.size, regions.ctypes.data, 3, out.ctypes.data, n) == 0
    exp = np.zeros(n, dtype=np.int32)
    exp[keep] = rdc
    assert np.array_equal(out, exp)
    assert hotlib.rsi_plot_expand(rdc.ctypes.data, rdc.size - 1, regions.ctypes.data, 3, out.ctypes.data, n) != 0   # lengths do not add up


def test_plot_files_of_a_deletion(hotlib, tmp_path):
    rng = np.random.default_rng(11)
    n = 60_000
    rd = rng.poisson(30, size=n).astype(np.int32)
    start, end = 30_000, 31_999
    rd[start - 1:end] = rng.poisson(15, size=end - start + 1)
    m, minmlen, chklen = 101, 3.01, 2.5
    dat, gp, img = (str(tmp_path / f"c.{e}") for e in ("dat", "gp", "ps"))
    c = _call(hotlib, start, end)
    hotlib.rsi_plot_write_files.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_double, C.c_double,
                                             C.c_char_p, C.c_double, C.c_char_p, C.c_char_p, C.c_char_p]
    rc = hotlib.rsi_plot_write_files(C.byref(c), b"chrS:30000~31999 2000 DEL", rd.ctypes.data, n, 30.0, m, minmlen, chklen, b"ps", 5.4,
                                     dat.encode(), gp.encode(), img.encode())
    assert rc == 0
    blocks = open(dat).read().split("\n\n\n")
    head, first = blocks[0].split("\n", 1)
    assert head.startswith(f"#{start} ~ {end}  {end - start + 1}  DEL")
    d = end - start + 1
    i1, i2 = int(start - chklen * d), int(end + chklen * d)
    # block 0: every position i1 .. start (step 1 here: fewer than 30000 positions), depth = rd[pos - 1]
    rows0 = [l.split("\t") for l in first.strip().split("\n")]
    assert [int(r[0]) for r in rows0] == list(range(i1, start + 1))
    assert [int(r[1]) for r in rows0] == [int(rd[p - 1]) for p in range(i1, start + 1)]
    # the running mean of the neighbourhood (the call cut out): width d + 1 (odd), ends filled with the first / last mean
    ref = np.concatenate([rd[i1 - 1:start - 1], rd[end:i2]]).astype(np.float64)
    band = d + ((d + 1) % 2)
    cs = np.concatenate([[0.0], np.cumsum(ref)])
    means = (cs[band:] - cs[:-band]) / band
    full = np.concatenate([np.full(band // 2, means[0]), means, np.full(len(ref) - len(means) - band // 2, means[-1])]).astype(np.float32)
    got = np.array([float(r[2]) for r in rows0[:-1]])
    assert np.allclose(got, full[:len(got)], rtol=2e-6)
    # block 1: the call itself
    rows1 = [l.split("\t") for l in blocks[1].strip().split("\n")]
    assert [int(r[0]) for r in rows1] == list(range(start, end + 1)) and all(r[2] == "NaN" for r in rows1)
    assert [int(r[1]) for r in rows1] == [int(v) for v in rd[start - 1:end]]
    # blocks 4-6: median / quartiles of the call; 7-9 of the neighbourhood
    lq, md, uq = _median_like(rd[start - 1:end])
    assert blocks[4].strip().split("\n")[0] == f"{start}\t{md:g}" and blocks[5].strip().split("\n")[0] == f"{start}\t{lq:g}"
    assert blocks[6].strip().split("\n")[1] == f"{end}\t{uq:g}"
    rlq, rmd, ruq = _median_like(ref.astype(np.int64))
    assert blocks[7].strip().split("\n")[0] == f"{i1}\t{rmd:g}" and blocks[9].strip().split("\n")[1] == f"{i2}\t{ruq:g}"
    script = open(gp).read()
    assert f'f="{dat}"' in script and 'set terminal postscript color enhanced solid' in script and f'set output "{img}"' in script
    assert "chrS:30000-31999 2000 DEL" in script and '30 w l lt 4 lw 4 t "CHROM med"' in script and script.rstrip().endswith("quit")
    # an old gnuplot gets the dialect without string variables
    hotlib.rsi_plot_write_files(C.byref(c), b"t", rd.ctypes.data, n, 30.0, m, minmlen, chklen, b"ps", 4.0, dat.encode(), gp.encode(), img.encode())
    assert open(gp).read().startswith("#f=")
    # a call beyond the array is refused (the reference draws an empty frame)
    assert hotlib.rsi_plot_write_files(C.byref(_call(hotlib, n + 5, n + 90)), b"t", rd.ctypes.data, n, 30.0, m, minmlen, chklen, b"ps", 5.0,
                                       dat.encode(), gp.encode(), img.encode()) != 0


GOLDEN_PLOTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plot_cases.npz")


@pytest.mark.parametrize("name", ["del_mid", "dup_wide", "del_short", "dup_left_edge", "del_right_edge"])
def test_plot_files_equal_the_references_own(hotlib, tmp_path, name, monkeypatch):
    """The writer against plot_icnv ITSELF (plotcnv.cpp:246-610): tests/golden/plot_cases.npz holds, per case, the per-base array,
    the call and the bytes the compiled reference wrote into its .dat and .gp file (tools/make_golden_plot.py).  Byte for byte:
    a deletion, a duplication wider than plot::pts positions (subsampled walk), a call shorter than m * minmlen, neighbourhoods
    clamped at either chromosome end.  The script is in the dialect the reference chose where the files were made
    (gnuplot_version() = -1: none installed)."""
    from rsicnv_amd import api
    g = np.load(GOLDEN_PLOTS, allow_pickle=False)
    rd = np.ascontiguousarray(g[name + "_rd"], dtype=np.int32)
    start, end, typ, length = (int(v) for v in g[name + "_call"])
    m, minmlen, chklen = (float(v) for v in g["params"])
    c = api.RsiCall()
    c.start, c.end, c.type, c.length, c.p1 = start, end, typ, length, float(g[name + "_p1"][0])
    base = str(g[name + "_base"])
    monkeypatch.chdir(tmp_path)                      # the data file's name is written into the script: same relative names
    os.makedirs("plots")
    hotlib.rsi_plot_write_files.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_double, C.c_int, C.c_double, C.c_double,
                                             C.c_char_p, C.c_double, C.c_char_p, C.c_char_p, C.c_char_p]
    rc = hotlib.rsi_plot_write_files(C.byref(c), str(g[name + "_title"]).encode(), rd.ctypes.data, rd.size, float(g[name + "_rdmed"][0]), int(m),
                                     minmlen, chklen, b"ps", float(g[name + "_version"][0]), (base + ".dat").encode(), (base + ".gp").encode(),
                                     (base + ".ps").encode())
    assert rc == 0
    assert open(base + ".dat").read() == str(g[name + "_dat"])
    assert open(base + ".gp").read() == str(g[name + "_gp"])
