"""GPU: generator twins, the worker pool, the command line against the reference's own binary,
error behaviour, and size-independent properties at a BASELINE.json size (60 Mb)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import calls_equal, make_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hot():
    from rsicnv_amd import api
    h = api.RsiHot(0)
    yield h
    h.close()


@pytest.mark.parametrize("model", [0, 1])
def test_device_generator_equals_host(hotlib, model):
    import torch
    from rsicnv_amd import synth
    plan = synth.make_plan(n=1_234_567, seed=0xFEED + model, model=model, n_events=7, gaps=2, max_len=30000, end_n=7000, gap_len=9000)
    fasta, depth = synth.generate_host(hotlib, plan)
    d_fa = torch.empty(plan["n"] + 64, dtype=torch.uint8, device="cuda")
    d_rd = torch.empty(plan["n"] + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(hotlib, plan, d_fa.data_ptr(), d_rd.data_ptr())
    assert np.array_equal(d_fa[:plan["n"]].cpu().numpy(), fasta)
    assert np.array_equal(d_rd[:plan["n"]].cpu().numpy(), depth)


def test_pool_equals_single_context(hot, hotlib):
    import torch
    from rsicnv_amd import api, synth
    plans = [synth.make_plan(n=300_000 + 50_017 * i, seed=0x9000 + i, model=i % 2, n_events=4, gaps=1, max_len=15000, end_n=4000,
                             gap_len=5000) for i in range(6)]
    bufs, chroms, single = [], [], []
    p = api.make_params()
    for pl in plans:
        d_fa = torch.empty(pl["n"] + 64, dtype=torch.uint8, device="cuda")
        d_rd = torch.empty(pl["n"] + 16, dtype=torch.int32, device="cuda")
        synth.generate_device(hotlib, pl, d_fa.data_ptr(), d_rd.data_ptr())
        bufs.append((d_fa, d_rd))
        chroms.append((d_rd.data_ptr(), d_fa.data_ptr(), pl["n"]))
        single.append(hot.run_device(p, d_rd.data_ptr(), d_fa.data_ptr(), pl["n"]))
    pool = api.RsiPool(0, 3)
    for _ in range(2):
        batch = pool.run(p, chroms)
        for a, b in zip(single, batch):
            ok, why = calls_equal(a.calls("calls_raw"), b.calls("calls_raw"), rtol=0)
            assert ok, why
            assert a.stats["RDmedian"] == b.stats["RDmedian"] and a.stats["RDsd"] == b.stats["RDsd"]
            # the fixed-size summary row of the cross-rank gather (rsi_result_summary), here with room for two calls only
            row = np.full(8 + 8 * 2, -7.0)
            calls = b.calls("calls")
            assert b.summary_into(row, 17, 2) == 8 + 8 * min(len(calls), 2)
            assert list(row[:5]) == [17.0, b.stats["RDmedian"], b.stats["RDsd"], len(calls), min(len(calls), 2)]
            for k, c in enumerate(calls[:2]):
                assert list(row[8 + 8 * k: 16 + 8 * k]) == [c["start"], c["end"], c["type"], c["qscore"], c["cnvmed"], c["cnviqr"], c["refmed"], c["refiqr"]]
            assert (row[8 + 8 * min(len(calls), 2):] == -7.0).all()
    # queued runs (rsi_pool_submit / rsi_pool_wait): three samples in the queue at once, different flags and chromosome subsets,
    # waited for out of order and from two threads; every run's results equal the single-context ones
    import threading
    p2 = api.make_params(m=51, trans=1)
    single2 = [hot.run_device(p2, c[0], c[1], c[2]) for c in chroms[:3]]
    for _ in range(3):
        h1 = pool.submit(p, chroms)
        h2 = pool.submit(p2, chroms[:3])
        h3 = pool.submit(p, chroms[::-1])
        got = {}
        th = threading.Thread(target=lambda: got.__setitem__("h3", pool.wait(h3)))
        th.start()
        got["h2"] = pool.wait(h2)
        got["h1"] = pool.wait(h1)
        th.join()
        for want, have in ((single, got["h1"]), (single2, got["h2"]), (single[::-1], got["h3"])):
            assert len(want) == len(have)
            for a, b in zip(want, have):
                ok, why = calls_equal(a.calls("calls_raw"), b.calls("calls_raw"), rtol=0)
                assert ok, why
                assert a.stats["RDmedian"] == b.stats["RDmedian"] and a.stats["RDsd"] == b.stats["RDsd"]
    empty = pool.submit(p, [])
    assert pool.wait(empty) == []
    with pytest.raises(api.RsiError):
        pool.wait({"ticket": 987654321, "k": 0, "out": None})
    pool.close()


def _write_case(tmp, fasta, depth, chrom="chrS"):
    fa = os.path.join(tmp, "ref.fa")
    with open(fa, "wb") as f:
        f.write(f">{chrom}\n".encode())
        seq = fasta.tobytes()
        for i in range(0, len(seq), 60):
            f.write(seq[i:i + 60] + b"\n")
    with open(fa + ".fai", "w") as f:
        f.write(f"{chrom}\t{len(fasta)}\t{len(chrom) + 2}\t60\t61\n")
    rd = os.path.join(tmp, "depth.txt")
    pos = np.arange(1, len(depth) + 1)
    with open(rd, "w") as f:
        f.write("#pos depth\n")
        np.savetxt(f, np.stack([pos, depth], axis=1), fmt="%d", delimiter="\t")
    return fa, rd


@pytest.mark.parametrize("extra", [[], ["-MED", "-m", "51"], ["-NOGC"]], ids=["nb", "med51", "nogc"])
def test_cli_output_file_matches_reference_binary(hotlib, tmp_path, extra):
    import oracle
    if not os.path.exists(oracle.REF_BIN):
        pytest.skip("oracle/_ref/rsicnv_ref not built")
    _, fasta, depth = make_case(hotlib, dict(n=400_007, seed=0xC11, model=1, n_events=5, gaps=1, max_len=20000, end_n=5000, gap_len=8000))
    fa, rd = _write_case(str(tmp_path), fasta, depth)
    ours, theirs = str(tmp_path / "ours.txt"), str(tmp_path / "ref.txt")
    exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
    subprocess.run([exe, "rsi", "-f", fa, "-d", rd, "-c", "chrS", "-o", ours, "-np"] + extra, check=True, capture_output=True, timeout=300)
    subprocess.run([oracle.REF_BIN, "rsi", "-f", fa, "-d", rd, "-c", "chrS", "-o", theirs, "-np"] + extra, check=True,
                   capture_output=True, timeout=600, cwd=str(tmp_path))
    a, b = open(ours, "rb").read(), open(theirs, "rb").read()
    assert a == b, f"output files differ:\n{a.decode()}\n---\n{b.decode()}"
    assert a.count(b"\n") >= 4
    # the per-L lines of the two rsistatus passes in OUT.log (rsi.cpp:1221-1224, 1251-1254): L, bins marked so far, bins, portion
    sweep = lambda path: [l for l in open(path + ".log").read().splitlines() if l.startswith(("DEL-\t", "DUP+\t"))]
    la, lb = sweep(ours), sweep(theirs)
    assert la == lb and len(la) >= 4 * 20, f"{len(la)} vs {len(lb)} sweep lines; first difference: " + \
        str(next(((x, y) for x, y in zip(la, lb) if x != y), None))
    # filterstatus' level table between the two passes (rsi.cpp:991-1002): level, bins, float mean of the level; then the two
    # chosen levels -- the same lines, in one piece, in the reference's log
    mine, ref = open(ours + ".log").read().splitlines(), open(theirs + ".log").read().splitlines()
    is_sweep = [l.startswith(("DEL-\t", "DUP+\t")) for l in mine]
    first = is_sweep.index(True)
    gap0 = next(i for i in range(first, len(mine)) if not is_sweep[i])          # first line behind the first pass
    gap1 = next(i for i in range(gap0, len(mine)) if is_sweep[i])               # the second pass begins
    table = mine[gap0:gap1]
    assert len(table) >= 3 and all(l.count("\t") in (1, 2) or l.startswith("warning") for l in table), table
    assert any(ref[i:i + len(table)] == table for i in range(len(ref))), "level table not in the reference log:\n" + "\n".join(table)
    if "-MED" not in extra:   # the NB transform's two lines (rsi.cpp:1140-1141)
        nbl = [l for l in mine if l.startswith("RD median")]
        assert len(nbl) == 2 and all(l in ref for l in nbl), nbl


def test_cli_plot_files(hotlib, tmp_path):
    """-p FOLDER -plotfiles: one data file and one gnuplot script per written call (plotcnv.cpp:613-672), each with the
    eleven data blocks plot_icnv writes, the call's block holding the capped, GC-adjusted depth at its positions."""
    _, fasta, depth = make_case(hotlib, dict(n=400_007, seed=0xC12, model=1, n_events=5, gaps=1, max_len=20000, end_n=5000, gap_len=8000))
    fa, rd = _write_case(str(tmp_path), fasta, depth)
    out, folder = str(tmp_path / "out.txt"), str(tmp_path / "plots")
    exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
    subprocess.run([exe, "rsi", "-f", fa, "-d", rd, "-c", "chrS", "-o", out, "-p", folder, "-plotfiles"], check=True, capture_output=True, timeout=300)
    rows = [l.split("\t") for l in open(out).read().splitlines() if not l.startswith("#")]
    assert len(rows) >= 3
    import shutil
    if shutil.which("gnuplot"):
        pytest.skip("gnuplot present: the files are piped through it and deleted, as in the reference")
    for r in rows:
        base = os.path.join(folder, f"rsi_chrS_{r[1]}_{r[2]}_{r[3]}")
        blocks = open(base + ".dat").read().split("\n\n\n")
        assert len(blocks) >= 11 and blocks[0].startswith(f"#{r[1]} ~ {r[2]}  {r[5]}  {r[3]}")
        body = [l.split("\t") for l in blocks[1].strip().split("\n")]
        # positions from START on, thinned to at most 30 000 points over the whole figure (plotcnv.cpp:382)
        assert int(body[0][0]) == int(r[1]) and int(r[2]) - 2 * max(1, 6 * int(r[5]) // 30000 + 1) <= int(body[-1][0]) <= int(r[2])
        assert all(x[2] == "NaN" for x in body)
        script = open(base + ".gp").read()
        assert f"chrS:{r[1]}-{r[2]} {r[5]} {r[3]}" in script and script.rstrip().endswith("quit")


def test_error_behaviour(hot, hotlib):
    from rsicnv_amd import api
    _, fasta, depth = make_case(hotlib, dict(n=300_000, seed=0xE44, model=0, n_events=3, gaps=0, max_len=9000, end_n=0))
    # chromosome too short for the 20-slice GC adjust: the reference exits (gccontent.cpp:66-71)
    with pytest.raises(api.RsiError) as e:
        hot.run(api.make_params(), depth[:4000], fasta[:4000])
    assert e.value.code == -4
    # negative depth is outside the contract of the integer histograms
    bad = depth.copy()
    bad[1234] = -3
    with pytest.raises(api.RsiError) as e:
        hot.run(api.make_params(), bad, fasta)
    assert e.value.code == -5
    # "Read depths too low, cannot call" (rsi.cpp:1809-1812): no calls, no error
    res = hot.run(api.make_params(), (depth // 10).astype(np.int32), fasta)
    assert res.stats["RDmedian"] < 5 and res.calls("calls_raw") == []
    # mismatched lengths are the caller's bug
    with pytest.raises(ValueError):
        hot.run(api.make_params(), depth[:-1], fasta)


@pytest.mark.parametrize("flags", [dict(trans=1, m=51, gcadjust=0), dict(trans=1, gcadjust=0), dict(trans=1)], ids=["med51_nogc", "med101_nogc", "med101"])
def test_equal_bin_medians_take_the_degenerate_quantile(hot, hotlib, oracle_cls, flags):
    """Every bin median the same: the 0.01-grid "median" of a selection that spans less than the grid step is its MEAN
    (partition_stat_tp's early return, wufunctions.cpp:371-381) -- the -MED transform's MAD is then 0, sigma 0, lambda the
    target alone.  Depth 28, 32, 30, 28, ... keeps the per-base MAD at 2 (the NB transform, always computed, stays finite)
    while every bin of 51 or 101 bases has median 30.  The product used to decline this input; it follows the reference now."""
    import oracle
    from rsicnv_amd import api
    _, fasta, _ = make_case(hotlib, dict(n=300_000, seed=0xF1A7, model=0, n_events=2, gaps=1, max_len=9000, end_n=3000, gap_len=5000))
    depth = np.tile(np.array([28, 32, 30], dtype=np.int32), fasta.size // 3 + 1)[:fasta.size].copy()
    depth[fasta == ord("N")] = 0
    O = oracle_cls()
    O.run(oracle.make_params(**flags), depth, fasta)
    res = hot.run(api.make_params(**flags), depth, fasta)
    sc = O.f64("scan_med")
    got = [res.stats[k] for k in ("tmedian1", "tsigma1", "tlamda1", "tmedian2", "tsigma2", "tlamda2")]
    assert got == list(sc[:6]) and got[0] == 30.0
    assert res.stats["Lmax"] == int(sc[7])
    assert np.array_equal(hot.fetch("binmedint"), O.i32("binmedint"))
    if not flags.get("gcadjust", 1):   # without the GC rescale every bin median IS 30: the first-pass MAD is the degenerate one
        assert set(hot.fetch("binmedint").tolist()) == {30} and got[1] == 0.0
    assert np.array_equal(hot.fetch("status2"), O.i32("med_status2"))
    ok, why = calls_equal(res.calls("calls"), O.calls("calls"))
    assert ok, why


def test_huge_depth_values_take_the_wide_gc_path(hot, hotlib, oracle_cls):
    """Depths of 2^21 and more invalidate the packed GC accumulators: the pipeline re-runs the wide
    form and must still match the oracle (the cap removes the outliers afterwards)."""
    import oracle
    from rsicnv_amd import api
    _, fasta, depth = make_case(hotlib, dict(n=400_003, seed=0xB16, model=0, n_events=4, gaps=1, max_len=15000, end_n=4000, gap_len=6000))
    depth = depth.copy()
    depth[[50_001, 123_457, 300_000, 399_990, 400_001]] = [3_000_000, 2_500_000, 70_000, 1 << 22, 99_999]
    O = oracle_cls()
    O.run(oracle.make_params(), depth, fasta)
    res = hot.run(api.make_params(), depth, fasta)
    assert res.stats["gc_rdmean"] == O.f64("chrom")[3]
    assert np.array_equal(hot.fetch("rd_gc"), O.i32("rd_gc"))
    assert np.array_equal(hot.fetch("rd_concat"), O.i32("rd_concat"))
    ok, why = calls_equal(res.calls("calls"), O.calls("calls"))
    assert ok, why


def test_properties_at_60mb(hot, hotlib):
    """BASELINE.json configs[1] size: checks that do not need the (slow) CPU path."""
    import torch
    from rsicnv_amd import api, synth
    plan = synth.config_plan(2)
    n = plan["n"]
    d_fa = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    d_rd = torch.empty(n + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(hotlib, plan, d_fa.data_ptr(), d_rd.data_ptr())
    p = api.make_params()
    r1 = hot.run_device(p, d_rd.data_ptr(), d_fa.data_ptr(), n)
    rdc = hot.fetch("rd_concat")
    medint = hot.fetch("binmedint")
    sums = hot.fetch("binsum")
    st2 = hot.fetch("status2")
    removed = int(np.sum(r1.noncode[1::2] - r1.noncode[0::2] + 1))
    m, nb = 101, len(medint)
    assert len(rdc) == n - removed and nb == len(rdc) // m
    bins = rdc[:nb * m].reshape(nb, m)
    assert np.array_equal(np.median(bins, axis=1).astype(np.int32), medint)      # exact order statistic, m odd
    assert np.array_equal(bins.sum(axis=1, dtype=np.int64), sums)                # checksum of checksums
    assert rdc.max() <= int(r1.stats["cap_median"] * 4.0)
    assert r1.stats["RDsd"] == pytest.approx(float(np.sqrt((rdc.astype(np.float64) ** 2).mean() - rdc.astype(np.float64).mean() ** 2)), rel=1e-12)
    # every implanted event is called with the right type, within two bins of its ends -- except the
    # last marked run of the chromosome, which the reference never emits
    calls = r1.calls("calls")
    for (a, b, code) in plan["events"][:-1]:
        typ = 0 if code in (1, 2) else 1
        hit = [c for c in calls if c["type"] == typ and abs(c["start"] - a) <= 2 * m and abs(c["end"] - b) <= 2 * m]
        assert hit, f"event {(a, b, code)} not called; calls: {[(c['start'], c['end'], c['type']) for c in calls]}"
    assert np.count_nonzero(st2) / nb < 0.2
    # idempotence: the inputs are not modified and a second run reproduces every call
    r2 = hot.run_device(p, d_rd.data_ptr(), d_fa.data_ptr(), n)
    ok, why = calls_equal(r1.calls("calls_raw"), r2.calls("calls_raw"), rtol=0)
    assert ok, why


def test_parity_20mb_vs_oracle(hot, hotlib, oracle_cls):
    import oracle
    from rsicnv_amd import api
    plan_kw = dict(n=20_000_011, seed=0x20B, model=1, n_events=24, gaps=3, max_len=90000, end_n=10000, gap_len=50000, centromere=300000)
    _, fasta, depth = make_case(hotlib, plan_kw)
    O = oracle_cls()
    O.run(oracle.make_params(), depth, fasta, snapshots=False)
    res = hot.run(api.make_params(), depth, fasta)
    assert np.array_equal(hot.fetch("rd_concat"), O.i32("rd_concat"))
    assert np.array_equal(hot.fetch("binmedint"), O.i32("binmedint"))
    assert np.array_equal(hot.fetch("status2"), O.i32("nb_status2"))
    for which in ("segs", "blocks", "calls_raw", "calls"):
        ok, why = calls_equal(res.calls(which), O.calls({"segs": "segs_nb"}.get(which, which)))
        assert ok, f"{which}: {why}"
    assert len(res.calls("calls")) >= 15



def test_host_candidate_path_still_agrees(hot, hotlib, oracle_cls, monkeypatch):
    """RSI_HOT_HOST_CANDIDATES=1 routes the per-base candidate steps through the host walks (the path a
    test takes when the device declines it): same calls as the oracle, and as the device path."""
    import oracle
    from rsicnv_amd import api
    plan_kw = dict(n=1_200_011, seed=0xCA4D, model=1, n_events=8, gaps=2, max_len=40000, end_n=6000, gap_len=20000)
    _, fasta, depth = make_case(hotlib, plan_kw)
    O = oracle_cls()
    O.run(oracle.make_params(), depth, fasta, snapshots=False)
    dev = hot.run(api.make_params(), depth, fasta)
    monkeypatch.setenv("RSI_HOT_HOST_CANDIDATES", "1")
    host = hot.run(api.make_params(), depth, fasta)
    monkeypatch.delenv("RSI_HOT_HOST_CANDIDATES")
    for which in ("blocks", "calls_raw", "calls"):
        ok, why = calls_equal(host.calls(which), O.calls(which))
        assert ok, f"host path {which}: {why}"
        ok, why = calls_equal(dev.calls(which), host.calls(which), rtol=1e-9)
        assert ok, f"device vs host path {which}: {why}"
    assert len(dev.calls("calls_raw")) >= 4


def _reference_text_semantics(lines, n):
    """load_data_from_text's loop (loaddata.cpp:496-517) in Python, for small inputs."""
    rd = np.zeros(n, dtype=np.int32)
    for ln in lines:
        if len(ln) < 1 or ln[0] == "#":
            continue
        toks = ln.split()
        def as_int(t):
            import re
            m = re.match(r"[+-]?\d+", t)
            return int(m.group(0)) if m else None
        pos = as_int(toks[0]) if toks else None
        if pos is None:
            continue                      # failed extraction leaves 0: skipped by pos < 1
        d = as_int(toks[1]) if len(toks) > 1 else None
        d = 0 if d is None else d
        if pos < 1:
            continue
        if pos >= n:
            break
        rd[pos - 1] = d
    return rd


def test_depth_text_parsed_on_device(hot, tmp_path):
    """Device text ingestion: comments, blank lines, CRLF, junk, missing positions, pos < 1,
    the stop at pos >= n -- against the reference loop's semantics; unsorted input takes the host loop."""
    rng = np.random.default_rng(7)
    n = 200_000
    depth = rng.poisson(30, n).astype(np.int32)
    lines = ["# header", ""]
    for pos in range(1, n + 40):                  # runs past the end: the loop must stop at pos >= n
        if pos % 997 == 0:
            continue                              # missing position stays 0
        d = int(depth[pos - 1]) if pos <= n else 77
        sep = "\t" if pos % 3 else "  "
        lines.append(f"{pos}{sep}{d}" + ("\r" if pos % 5 == 0 else ""))
        if pos == 1000:
            lines += ["#comment in the middle", "", "chrS\t5\t9", "0\t55", "-3 4"]   # all skipped
        if pos == 2000:
            lines.append(f"{pos + 1}")            # position without a depth: stored as 0 ... then overwritten? no: next line is pos+1 again
    text = "\n".join(lines)                       # no trailing newline
    # the duplicate of position 2001 makes this file unsorted -> host loop; build a sorted variant as well
    p_uns = tmp_path / "unsorted.txt"; p_uns.write_text(text)
    sorted_lines = [ln for i, ln in enumerate(lines) if not (ln == "2001")]
    p_srt = tmp_path / "sorted.txt"; p_srt.write_text("\n".join(sorted_lines) + "\n")
    for path, src, want_fallback in ((p_srt, sorted_lines, 0), (p_uns, lines, 1)):
        st = hot.load_depth_text(str(path), n)
        got = hot.fetch("depth_in")
        exp = _reference_text_semantics(src, n)
        assert st["fallback"] == want_fallback, st
        assert np.array_equal(got, exp), (path.name, int(np.sum(got != exp)))
        assert st["bytes"] == path.stat().st_size and st["stored"] > 0.99 * n * (1 - 1 / 997) - 10


def test_depth_text_big_file_matches_arrays(hot, hotlib, tmp_path):
    """A 3 Mb chromosome through text (several 64 MB chunks would need > 5 Mb; this one checks the run
    end to end): same calls as from the arrays."""
    from rsicnv_amd import api
    plan_kw = dict(n=3_000_017, seed=0x7E47, model=1, n_events=10, gaps=2, max_len=60000, end_n=8000, gap_len=30000)
    _, fasta, depth = make_case(hotlib, plan_kw)
    path = tmp_path / "depth.txt"
    n = depth.size
    pos = np.arange(1, n + 1)
    with open(path, "w") as f:
        np.savetxt(f, np.column_stack([pos, depth]), fmt="%d", delimiter="\t")
    r_arr = hot.run(api.make_params(), np.concatenate([depth[:-1], [0]]).astype(np.int32), fasta)   # the last base is never set (Q7)
    r_txt = hot.run_text(api.make_params(), str(path), fasta)
    assert r_txt.text_stats["fallback"] == 0 and r_txt.text_stats["lines"] == n
    ok, why = calls_equal(r_txt.calls("calls_raw"), r_arr.calls("calls_raw"), rtol=0)
    assert ok, why


@pytest.mark.parametrize("flag_kw", [dict(epsilon=2.5, chklen=1.5, merge=0), dict(trans=1, threshold=0.8, maxchkbp=2000, m=75),
                                     dict(epsilon=1.0, cap=2.5, m=201), dict(trans=2, chklen=4.0, minmlen=5.0, buffer=0.2)],
                         ids=["eps_reflen_nomerge", "med_threshold_maxchkbp_m75", "eps1_cap25_m201", "all_reflen4_minmlen5_buffer02"])
def test_less_common_flags_parity(hot, hotlib, oracle_cls, flag_kw):
    """-e, -threshold, -reflen, -maxchkbp, -nomerge and unusual bin sizes: status arrays and calls against the oracle
    (which test_oracle_vs_ref pins against the compiled reference for the first two combinations)."""
    import oracle
    from rsicnv_amd import api
    plan_kw = dict(n=2_000_003, seed=0xF1A6 + len(flag_kw), model=1, n_events=9, gaps=2, max_len=60000, end_n=10000, gap_len=30000)
    _, fasta, depth = make_case(hotlib, plan_kw)
    O = oracle_cls()
    O.run(oracle.make_params(**flag_kw), depth, fasta)
    res = hot.run(api.make_params(**flag_kw), depth, fasta)
    trans = flag_kw.get("trans", 0)
    pre = "med" if trans == 1 else "nb"
    assert res.stats["Lmax"] == int(O.f64(f"scan_{pre}")[7])
    for w in ("status1", "status1f", "status2"):
        assert np.array_equal(hot.fetch(w), O.i32(f"{pre}_{w}")), w
    segs_o = (O.calls("segs_med") if trans != 0 else []) + (O.calls("segs_nb") if trans != 1 else [])
    for which, exp in (("segs", segs_o), ("blocks", O.calls("blocks")), ("calls_raw", O.calls("calls_raw")), ("calls", O.calls("calls"))):
        ok, why = calls_equal(res.calls(which), exp)
        assert ok, f"{which}: {why}"


def test_bin_size_extremes(hot, hotlib, oracle_cls):
    """m = 1001 (bin medians from LDS, the kernel's general path) against the oracle; m = 3 asks for a scan length of
    3333 windows, beyond the kernel's 2048: a clean RSI_ERR_UNSUPPORTED, not a wrong answer."""
    import oracle
    from rsicnv_amd import api
    plan_kw = dict(n=3_000_001, seed=0xB161, model=0, n_events=6, gaps=1, max_len=120000, end_n=10000, gap_len=30000)
    _, fasta, depth = make_case(hotlib, plan_kw)
    O = oracle_cls()
    O.run(oracle.make_params(m=1001), depth, fasta)
    res = hot.run(api.make_params(m=1001), depth, fasta)
    assert np.array_equal(hot.fetch("binmedint"), O.i32("binmedint"))
    assert np.array_equal(hot.fetch("status2"), O.i32("nb_status2"))
    for which in ("blocks", "calls_raw", "calls"):
        ok, why = calls_equal(res.calls(which), O.calls(which))
        assert ok, f"{which}: {why}"
    # bins of 1501 at four times the depth: the cap (4 x 120) needs the 256-value LDS histogram (32 KB) next to a 48 KB
    # tile, more than the 64 KB a kernel gets without asking for it
    deep = (depth * 4 + (np.arange(depth.size, dtype=np.int64) * 2654435761 % 4).astype(np.int32) * (depth > 0)).astype(np.int32)
    O.run(oracle.make_params(m=1501), deep, fasta)
    res = hot.run(api.make_params(m=1501), deep, fasta)
    assert res.stats["RDmedian"] > 100
    assert np.array_equal(hot.fetch("binmedint"), O.i32("binmedint"))
    assert np.array_equal(hot.fetch("rd_concat"), O.i32("rd_concat"))
    ok, why = calls_equal(res.calls("calls"), O.calls("calls"))
    assert ok, why
    # -m 3 (Lmax 3333) fits the scan's LDS tile (golden case wide_m3_nb); -m 1 (Lmax 10000) runs on tiles in device memory
    # (test_scan_of_ten_thousand_lengths); a chromosome with fewer bins than the scan is long is refused as the reference refuses it
    with pytest.raises(api.RsiError) as e:
        hot.run(api.make_params(m=1), depth[:9_000], fasta[:9_000])
    assert e.value.code != 0


def test_scan_of_ten_thousand_lengths(hot, oracle_cls):
    """-m 1: Lmax = 10000 (rsi.cpp:1830), beyond what an LDS tile of the scan kernel holds -- the exact sweep runs on tiles in
    device memory behind the detection pass.  At that bin size the stages BEHIND the scan take the reference, the oracle and
    this library's host side hours for any chromosome with more bins than the scan is long (the oracle: 400 s for 5500 bins),
    so the scan pass is driven alone (rsi_hot_debug_scan) over synthetic bins, and its status array compared with a
    restatement of rsistatus that advances all window lengths together (tests/scan_restatement.py, itself checked against the
    oracle on the CPU, test_golden_oracle.py)."""
    from scan_restatement import long_scan_case, rsistatus_numpy
    T, medint, RDmedian, tmedian, tlamda, Lmax = long_scan_case()
    exp = rsistatus_numpy(T, medint, RDmedian, tmedian, tlamda, Lmax, oracle_cls().exact_median)
    assert np.count_nonzero(exp == -4980) > 4000 and (exp[3000:3040] < 0).all() and (exp[7000:7050] > 0).all()
    got, info = hot.debug_scan(T, medint, RDmedian, tmedian, tlamda, Lmax)
    assert np.array_equal(got, exp), f"{np.count_nonzero(got != exp)} bins differ; first at {np.nonzero(got != exp)[0][:5]}"
    assert info[0] > 0 and info[1] == 0 and info[2] == 0     # tiles listed by the detection pass; no escaped walk, no inexact threshold
    # the same bins at a length an LDS tile holds: the two sweeps agree with the restatement there too
    exp3 = rsistatus_numpy(T, medint, RDmedian, tmedian, tlamda, 3333, oracle_cls().exact_median)
    got3, _ = hot.debug_scan(T, medint, RDmedian, tmedian, tlamda, 3333)
    assert np.array_equal(got3, exp3) and not np.array_equal(exp3, exp)


@pytest.mark.parametrize("nb,Lmax,plateau,quiet", [
    (44_000, 12_000, (17_000, 28_000), (11_000, 38_000)),     # device-memory tiles behind the detection pass, beyond the resident work block
    (64_000, 24_000, (20_000, 43_000), (11_000, 58_000)),     # no detection pass (its LDS stretch ends near 13 000), per-L counts in device memory
    (80_000, 33_000, (22_000, 54_000), (11_000, 74_000)),     # 32-bit staged indices (the staged stretch passes 32 768 bins)
], ids=["L12000", "L24000", "L33000"])
@pytest.mark.timeout(1500)
def test_scan_beyond_the_resident_length(hot, oracle_cls, nb, Lmax, plateau, quiet):
    """a scan longer than 10 400 lengths (the reference takes Lmax = max(10000 / m, cal_max), rsi.cpp:1286-1289,
    1830-1831: a large -threshold or a very noisy chromosome computes such a length) used to be refused.  The scan pass alone
    (rsi_hot_debug_scan) against the restatement of rsistatus, at three lengths that each switch one more piece of the long form on."""
    from scan_restatement import long_scan_case, rsistatus_numpy
    T, medint, RDmedian, tmedian, tlamda, Lmax = long_scan_case(nb, Lmax, plateau, quiet)
    exp = rsistatus_numpy(T, medint, RDmedian, tmedian, tlamda, Lmax, oracle_cls().exact_median)
    width = plateau[1] - plateau[0]
    first_len = -int(exp[plateau[0] + width // 2])       # the plateau is first seen by windows of about its own length (float rounding moves it a bin or two)
    assert abs(first_len - (width - 20)) <= 3 and np.count_nonzero(exp == -first_len) > 4000 and (exp[3000:3040] < 0).all() and (exp[7000:7050] > 0).all()
    got, info = hot.debug_scan(T, medint, RDmedian, tmedian, tlamda, Lmax)
    assert np.array_equal(got, exp), f"{np.count_nonzero(got != exp)} bins differ; first at {np.nonzero(got != exp)[0][:5]}"
    assert info[1] == 0 and info[2] == 0
    assert (info[0] > 0) == (Lmax <= 13_000)      # the detection pass listed tiles where it ran


@pytest.mark.parametrize("threshold,n,lmax", [(27.0, 1_450_001, 11_663), (46.0, 4_100_001, 33_854)], ids=["cal_max_11663", "cal_max_33854"])
@pytest.mark.timeout(1500)
def test_whole_run_with_a_computed_scan_length_beyond_10400(hot, hotlib, oracle_cls, threshold, n, lmax):
    """VERDICT r4 item 8, the whole path: -MED with a large -threshold computes cal_max = (4 threshold)^2 (rsi.cpp:1432-1433) -- 11 663 and 33 854 lengths -- which the library used to refuse.  The events are short (30 bins of no
    coverage, 40 bins at 1.5 x): the second pass, which drops -threshold (rsi.cpp:1457-1469) but keeps Lmax, then hits at a
    few hundred lengths only; an event of a thousand bins would hit at every length up to Lmax and cost the reference (and
    the oracle) hours in window medians.  Status arrays, scan parameters and calls against the oracle."""
    import oracle
    from rsicnv_amd import api, synth
    m = 101
    plan = synth.make_plan(n=n, seed=0x5CB1, model=0, mean=40.0, n_events=0, gaps=0, max_len=1000, end_n=1000)
    fasta, depth = synth.generate_host(hotlib, plan)
    depth = depth.copy()
    depth[400_000:400_000 + 30 * m] = 0
    depth[1_000_000:1_000_000 + 40 * m] = (depth[1_000_000:1_000_000 + 40 * m] * 3) // 2
    flags = dict(m=m, trans=1, threshold=threshold)
    O = oracle_cls()
    O.run(oracle.make_params(**flags), depth, fasta)
    sc = O.f64("scan_med")
    assert int(sc[7]) == lmax
    res = hot.run(api.make_params(**flags), depth, fasta)
    st = res.stats
    assert st["Lmax"] == lmax and "scan.long (Lmax > 10400)" in dict(hot.phase_times())
    np.testing.assert_allclose([st[k] for k in ("tmedian1", "tsigma1", "tlamda1", "tmedian2", "tsigma2", "tlamda2")], sc[:6], rtol=1e-12)
    assert st["trim_escapes"] == int(sc[9]) and st["inexact_sums"] == 0
    for mine, theirs in (("status1", "med_status1"), ("status1f", "med_status1f"), ("status2", "med_status2")):
        assert np.array_equal(hot.fetch(mine), O.i32(theirs)), mine
    assert np.count_nonzero(O.i32("med_status2")) > 50
    for which, exp in (("segs", O.calls("segs_med")), ("calls_raw", O.calls("calls_raw")), ("calls", O.calls("calls"))):
        ok, why = calls_equal(res.calls(which), exp)
        assert ok, f"{which}: {why}"
    assert len(res.calls("calls")) >= 1


def test_many_n_runs(hot, hotlib, oracle_cls):
    """About a thousand N runs (2000 run boundaries: more than the 1024 the first round trip brings back): regions,
    compaction and calls against the oracle."""
    import oracle
    from rsicnv_amd import api
    plan_kw = dict(n=6_000_003, seed=0x4E4E, model=0, n_events=6, gaps=2000, max_len=30000, end_n=3000, gap_len=120)
    _, fasta, depth = make_case(hotlib, plan_kw)
    O = oracle_cls()
    O.run(oracle.make_params(), depth, fasta)
    res = hot.run(api.make_params(), depth, fasta)
    assert len(res.noncode) // 2 > 512
    assert np.array_equal(res.noncode, O.i32("noncode"))
    assert np.array_equal(hot.fetch("rd_concat"), O.i32("rd_concat"))
    assert np.array_equal(hot.fetch("status2"), O.i32("nb_status2"))
    for which in ("calls_raw", "calls"):
        ok, why = calls_equal(res.calls(which), O.calls(which))
        assert ok, f"{which}: {why}"


@pytest.mark.parametrize("flags", [dict(), dict(gcadjust=0), dict(cap=-1.0), dict(m=51, trans=1)], ids=["default", "nogc", "nocap", "m51_med"])
def test_median_depth_beyond_65535(hot, hotlib, oracle_cls, flags):
    """a chromosome whose median depth lies above the 65 536 values of the device's integer histograms (here
    about 120 000) used to be refused.  The cap median (loaddata.cpp:233), the chromosome median / SD (rsi.cpp:2202-2203) and
    the MAD subsamples (rsi.cpp:1127-1143) then come from the arrays themselves on the host; everything else takes the
    deep-coverage kernels.  All stages against the oracle."""
    import oracle
    from rsicnv_amd import api
    _, fasta, depth = make_case(hotlib, dict(n=1_200_007, seed=0xDEEA, model=0, n_events=6, gaps=1, max_len=40000, end_n=4000, gap_len=9000))
    fill = np.random.default_rng(0xDEEA).integers(0, 4000, size=depth.size, dtype=np.int32)
    depth = np.where(depth > 0, depth * 4000 + fill, 0).astype(np.int32)
    assert np.median(depth[depth > 0]) > 100_000
    O = oracle_cls()
    O.run(oracle.make_params(**flags), depth, fasta)
    res = hot.run(api.make_params(**flags), depth, fasta)
    st = res.stats
    phases = dict(hot.phase_times())
    assert "a6.statistics on the host (depth > 65535)" in phases
    assert ("a4.median on the host (depth > 65535)" in phases) == (flags.get("cap", 4.0) > 1)
    if flags.get("gcadjust", 1):
        assert np.array_equal(hot.fetch("rd_gc"), O.i32("rd_gc"))
    assert np.array_equal(hot.fetch("rd_concat"), O.i32("rd_concat"))
    assert st["RDmedian"] == O.f64("chrom")[0] and st["RDsd"] == O.f64("chrom")[1]
    if flags.get("cap", 4.0) > 1:
        assert st["cap_median"] == O.f64("chrom")[2] > 65535
    assert np.array_equal(hot.fetch("binmedint"), O.i32("binmedint"))
    nbs = O.f64("nb")
    assert st["nb_mad"] == nbs[1] and st["nb_r"] == nbs[2]
    pre = "med" if flags.get("trans", 0) == 1 else "nb"
    for mine, theirs in (("status1", f"{pre}_status1"), ("status1f", f"{pre}_status1f"), ("status2", f"{pre}_status2")):
        assert np.array_equal(hot.fetch(mine), O.i32(theirs)), mine
    for which in ("blocks", "calls_raw", "calls"):
        ok, why = calls_equal(res.calls(which), O.calls(which))
        assert ok, f"{which}: {why}"
    assert len(res.calls("calls")) >= 1


@pytest.mark.gpu
def test_deep_coverage_sends_tests_to_the_host_walk(hot, hotlib, oracle_cls):
    """At 12000x the values of a candidate span more integer buckets than the device
    histogram holds: the kernel declines those tests (CandOut.flags) and the host walk serves them
    from pages of the device depth.  The mix of device and host tests must still match the oracle."""
    import oracle
    from rsicnv_amd import api
    _, fasta, depth = make_case(hotlib, dict(n=1_500_003, seed=0xDEE9, model=0, n_events=7, gaps=1,
                                             max_len=40000, end_n=4000, gap_len=9000))
    # the generator's sampling tables stop at a few hundred: scale a 30x track up and fill the comb
    fill = np.random.default_rng(0xDEE9).integers(0, 400, size=depth.size, dtype=np.int32)
    depth = np.where(depth > 0, depth * 400 + fill, 0).astype(np.int32)
    O = oracle_cls()
    O.run(oracle.make_params(), depth, fasta)
    res = hot.run(api.make_params(), depth, fasta)
    phases = dict(hot.phase_times())
    assert phases["calls.host_fallbacks"] > 0, phases
    ok, why = calls_equal(res.calls("calls"), O.calls("calls"))
    assert ok, why
    ok, why = calls_equal(res.calls("calls_raw"), O.calls("calls_raw"))
    assert ok, why
    assert len(res.calls("calls")) > 0


def test_gc_level_with_a_tiny_mean_rescales_past_the_lds_histograms(hot, hotlib, oracle_cls):
    """A soft-masked stretch shares GC level 0 with the N runs; give it a depth of 8 with a sprinkle of 200s and the level's mean
    is ~1.7, its bases are rescaled by ~16: values in the thousands from BYTE depths -- past K3''s 512-value LDS histogram
    (global atomics), saturated in the byte copy, capped by K4'.  Everything must still equal the oracle's arrays."""
    import oracle
    from rsicnv_amd import api
    plan, fasta, depth = make_case(hotlib, dict(n=1_600_003, seed=0x7199, model=1, n_events=6, gaps=1, max_len=30000, end_n=4000, gap_len=9000))
    (a, b), = [tuple(x) for x in plan["lower"]][:1]
    depth = depth.copy()
    depth[a:b] = 8
    depth[a:b:97] = 200
    O = oracle_cls()
    O.run(oracle.make_params(), depth, fasta)
    res = hot.run(api.make_params(), depth, fasta)
    rd_gc = hot.fetch("rd_gc")
    assert rd_gc.max() >= 512, rd_gc.max()              # the case does what it says
    assert np.array_equal(rd_gc, O.i32("rd_gc"))
    assert np.array_equal(hot.fetch("rd_concat"), O.i32("rd_concat"))
    assert res.stats["RDmedian"] == O.f64("chrom")[0]
    ok, why = calls_equal(res.calls("calls"), O.calls("calls"))
    assert ok, why


@pytest.mark.parametrize("model", [0, 1])
def test_300x_coverage_takes_the_int32_kernels(hot, hotlib, oracle_cls, model):
    """At 300x hardly a base fits K2's byte copy: K3' hands over, the int32 rescale (LDS histogram anchored at the mean depth)
    and the int32 K4 (anchored at the cap median) run instead.  Same bars as everywhere: arrays bit for bit, calls equal."""
    import oracle
    from rsicnv_amd import api
    _, fasta, depth = make_case(hotlib, dict(n=2_000_003, seed=0x300 + model, model=model, n_events=8, gaps=1,
                                             max_len=40000, end_n=4000, gap_len=9000))
    fill = np.random.default_rng(0x300).integers(0, 10, size=depth.size, dtype=np.int32)
    depth = np.where(depth > 0, depth * 10 + fill, 0).astype(np.int32)
    O = oracle_cls()
    O.run(oracle.make_params(), depth, fasta)
    res = hot.run(api.make_params(), depth, fasta)
    assert res.stats["byte_escapes"] > depth.size // 8
    assert "a2-3.deep coverage" in dict(hot.phase_times())
    assert np.array_equal(hot.fetch("rd_gc"), O.i32("rd_gc"))
    assert np.array_equal(hot.fetch("rd_concat"), O.i32("rd_concat"))
    assert np.array_equal(hot.fetch("binmedint"), O.i32("binmedint"))
    assert res.stats["RDmedian"] == O.f64("chrom")[0] and res.stats["RDsd"] == pytest.approx(O.f64("chrom")[1], rel=1e-12)
    ok, why = calls_equal(res.calls("calls_raw"), O.calls("calls_raw"))
    assert ok, why
    ok, why = calls_equal(res.calls("calls"), O.calls("calls"))
    assert ok, why
    assert len(res.calls("calls")) > 0


def test_300x_coverage_is_not_a_cliff(hot, hotlib):
    """The same 60 Mb chromosome at 30x and scaled to 300x (depth * 10 + comb): the deep path streams 4-byte intermediates and
    runs one more pass, so it is slower -- but by a small factor, not by the order of magnitude that value histograms fed with
    global atomics cost before their LDS windows followed the depth."""
    import time
    import torch
    from rsicnv_amd import api, synth
    plan = synth.config_plan(2)
    n = plan["n"]
    d_fa = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    d_rd = torch.empty(n + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(hotlib, plan, d_fa.data_ptr(), d_rd.data_ptr())
    g = torch.Generator(device="cuda").manual_seed(300)
    d_deep = torch.where(d_rd > 0, d_rd * 10 + torch.randint(0, 10, d_rd.shape, device="cuda", dtype=torch.int32, generator=g), torch.zeros_like(d_rd))
    p = api.make_params()

    def best_of(buf, k=4):
        ts = []
        for _ in range(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = hot.run_device(p, buf.data_ptr(), d_fa.data_ptr(), n)
            ts.append(time.perf_counter() - t0)
        return min(ts[1:]), r
    t30, r30 = best_of(d_rd)
    t300, r300 = best_of(d_deep)
    assert r300.stats["byte_escapes"] > n // 8 and r30.stats["byte_escapes"] == 0
    assert r300.stats["RDmedian"] > 250
    calls30 = [(c["start"], c["end"], c["type"]) for c in r30.calls("calls")]
    calls300 = [(c["start"], c["end"], c["type"]) for c in r300.calls("calls")]
    assert len(calls300) >= len(calls30) - 1          # the same events are there at ten times the depth
    print(f"60 Mb: {t30*1e3:.2f} ms at 30x, {t300*1e3:.2f} ms at 300x")
    assert t300 < 3.0 * t30, (t30, t300)


def _level_sums_numpy(T, status, Lmax):
    """The reference's loop (rsi.cpp:967-976): per status level, a float accumulation in index order (np.cumsum is sequential)."""
    sums = np.zeros(2 * Lmax + 1, dtype=np.float32)
    counts = np.zeros(2 * Lmax + 1, dtype=np.int32)
    for l in np.unique(status):
        x = T[status == l].astype(np.float32)
        sums[l + Lmax] = np.cumsum(x, dtype=np.float32)[-1]
        counts[l + Lmax] = x.size
    return sums, counts


@pytest.mark.parametrize("case", ["nb_like", "magnitudes", "ties", "edges"])
def test_filterstatus_level_sums_are_the_sequential_float_sums(hot, case):
    """The device's parallel form of filterstatus' per-level float accumulation (kernels_fs.hip) against the sequential loop, bit
    for bit: values of one magnitude (the real case: 25 binade crossings in 2.5 M bins), of wildly mixed magnitudes (a crossing
    every few bins), values that tie exactly at every rounding (powers of two), and the corner shapes."""
    rng = np.random.default_rng({"nb_like": 1, "magnitudes": 2, "ties": 3, "edges": 4}[case])
    Lmax = 99
    def marks(nb, frac):
        st = np.zeros(nb, dtype=np.int32)
        k = int(nb * frac)
        idx = rng.choice(nb, size=k, replace=False)
        st[idx] = rng.integers(1, Lmax + 1, size=k) * rng.choice([-1, 1], size=k)
        return st
    inputs = []
    if case == "nb_like":
        nb = 2_470_000
        inputs.append((rng.gamma(90.0, 1.0 / 3.0, nb).astype(np.float32), marks(nb, 0.01)))
        inputs.append((np.round(rng.gamma(30.0, 1.0, nb)).astype(np.float32), marks(nb, 0.05)))      # -MED: integer-valued floats
    elif case == "magnitudes":
        nb = 600_000
        inputs.append(((10.0 ** rng.uniform(-2, 4, nb)).astype(np.float32), marks(nb, 0.02)))
        t = (10.0 ** rng.uniform(-3, 1, nb)).astype(np.float32); t[rng.random(nb) < 0.3] = 0.0
        inputs.append((t, marks(nb, 0.3)))
    elif case == "ties":
        nb = 1_500_000
        inputs.append((rng.choice(np.array([0.5, 1, 2, 4, 8, 16, 3, 6, 12, 24, 1.5], dtype=np.float32), nb), marks(nb, 0.01)))
        inputs.append((np.full(nb, 4.0, dtype=np.float32), np.zeros(nb, dtype=np.int32)))           # every add a tie from 2^26 on
    else:
        for nb in (1, 7, 2047, 2048, 2049, 5000):
            inputs.append((rng.gamma(9.0, 3.0, nb).astype(np.float32), marks(nb, 0.2)))
        nb = 10_000
        inputs.append((rng.gamma(9.0, 3.0, nb).astype(np.float32), rng.integers(1, 5, nb).astype(np.int32)))   # nothing unmarked
        st = np.zeros(nb, dtype=np.int32); st[:4000] = -3                                                       # the sum starts late
        inputs.append((rng.gamma(9.0, 3.0, nb).astype(np.float32), st))
    for T, st in inputs:
        got_s, got_c = hot.debug_level_sums(T, st, Lmax)
        exp_s, exp_c = _level_sums_numpy(T, st, Lmax)
        assert np.array_equal(got_c, exp_c)
        assert np.array_equal(got_s.view(np.uint32), exp_s.view(np.uint32)), \
            [(l - Lmax, float(a), float(b)) for l, (a, b) in enumerate(zip(got_s, exp_s)) if a != b][:5]
    # a negative value: the integer-step form does not apply, the device says so and the pipeline runs the loop itself
    T = rng.gamma(9.0, 3.0, 5000).astype(np.float32); T[1234] = -1.0
    _, c = hot.debug_level_sums(T, np.zeros(5000, dtype=np.int32), Lmax)
    assert c[Lmax] == -1


@pytest.mark.gpu
def test_depth_narrowed_on_the_host_arrives_unchanged(hot, hotlib, monkeypatch):
    """rsi_hot_run's narrowed upload (the depth crosses PCIe as bytes, the values of 255 and more as a list; the device widens
    and patches): what reaches the device is the caller's int32 array, value for value -- including a handful of deep and of
    absurd values -- and the run's results are those of the plain upload (the reference's golden file).  More escapes than the
    list holds: the plain upload, same results."""
    import golden_util as gu
    from conftest import make_case
    from rsicnv_amd import api
    monkeypatch.setenv("RSI_HOT_H2D_NARROW", "1")
    for _ in range(2):
        gu.check_hip_against_golden(hot, hotlib, "gampois_nb_m101")
    _, fasta, depth = make_case(hotlib, dict(n=700_000, seed=0xD8, model=1, n_events=4, gaps=1, max_len=15000, end_n=4000, gap_len=5000))
    depth = depth.copy()
    rng = np.random.default_rng(5)
    pos = rng.choice(np.arange(20_000, 680_000), size=300, replace=False)
    depth[pos[:100]] = 255
    depth[pos[100:200]] = rng.integers(256, 4000, size=100)
    depth[pos[200:]] = rng.integers(70_000, 2_000_000, size=100)
    res_n = hot.run(api.make_params(), depth, fasta)
    assert np.array_equal(hot.fetch("depth_in"), depth)
    monkeypatch.setenv("RSI_HOT_H2D_SPLIT_MIN", "300000")      # the two-thread form a long chromosome takes
    hot.run(api.make_params(), depth, fasta)
    assert np.array_equal(hot.fetch("depth_in"), depth)
    monkeypatch.delenv("RSI_HOT_H2D_SPLIT_MIN")
    calls_n = [(c["start"], c["end"], c["type"], c["score"]) for c in res_n.calls("calls")]
    stats_n = (res_n.stats["RDmedian"], res_n.stats["RDsd"], res_n.stats["byte_escapes"])
    monkeypatch.setenv("RSI_HOT_H2D_NARROW", "0")
    res_p = hot.run(api.make_params(), depth, fasta)
    assert np.array_equal(hot.fetch("depth_in"), depth)
    assert calls_n == [(c["start"], c["end"], c["type"], c["score"]) for c in res_p.calls("calls")]
    assert stats_n == (res_p.stats["RDmedian"], res_p.stats["RDsd"], res_p.stats["byte_escapes"]) and stats_n[2] == 300
    # more values of 255 and more than the list holds (n / 64): the plain upload
    monkeypatch.setenv("RSI_HOT_H2D_NARROW", "1")
    deep = depth.copy()
    deep[rng.choice(700_000, size=30_000, replace=False)] = 300
    hot.run(api.make_params(), deep, fasta)
    assert np.array_equal(hot.fetch("depth_in"), deep)
