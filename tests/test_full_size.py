"""GPU: every BASELINE.json configuration at its STATED size (the parity matrix of test_hot_parity.py runs the same
workloads at sizes the CPU oracle finishes in seconds).  Checks that scale: call tables against golden files made by the
compiled reference at full size (tools/make_golden_full.py), size-independent properties of the per-base and bin arrays
(checksum of checksums, exact bin medians, cap, moments), the 24 chromosomes in flight against one at a time."""
import json
import os

import numpy as np
import pytest

import golden_util as gu
from conftest import calls_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hot():
    from rsicnv_amd import api
    h = api.RsiHot(0)
    yield h
    h.close()


def _device_case(lib, plan):
    import torch
    from rsicnv_amd import synth
    n = plan["n"]
    d_fa = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    d_rd = torch.empty(n + 16, dtype=torch.int32, device="cuda")
    synth.generate_device(lib, plan, d_fa.data_ptr(), d_rd.data_ptr())
    torch.cuda.synchronize()
    return d_rd, d_fa


def _golden(name):
    path = os.path.join(gu.GOLDEN_DIR, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated (tools/make_golden_full.py needs the compiled reference)")
    g = np.load(path, allow_pickle=False)
    return g, json.loads(str(g["plan"])), json.loads(str(g["flags"]))


def _plan(plan):
    plan = dict(plan)
    for k in ("events", "nruns", "lower"):
        plan[k] = [tuple(e) for e in plan[k]]
    return plan


def _check_against_golden(res, g, what, chrom=None):
    assert np.array_equal(res.noncode, g["noncode"]), what
    assert (res.stats["RDmedian"], res.stats["RDsd"]) == tuple(g["chrom_scalars"]), what
    assert res.stats["n_compact"] == int(g["n_compact"])
    for which in ("calls_raw", "calls"):
        ok, why = calls_equal(res.calls(which), gu.calls_from_array(g[which]))
        assert ok, f"{what} {which}: {why}"
    if chrom is not None and "rows" in g:   # the output rows as the reference's cnv_format1 printed them (rsi.cpp:581-631)
        assert "\n".join(res.format_rows(chrom)) + ("\n" if res.rows else "") == str(g["rows"]), what


def _genome_hash(results):
    """bench.py's step hash: every row + newline in chromosome order, then (chromosome index, RDmedian, RDsd) per chromosome."""
    import hashlib
    h = hashlib.sha256()
    for c, r in enumerate(results):
        for row in r.format_rows(f"chr{c + 1}"):
            h.update(row.encode()); h.update(b"\n")
    for c, r in enumerate(results):
        h.update(repr((c, float(r.stats["RDmedian"]), float(r.stats["RDsd"]))).encode())
    return h.hexdigest()


def _genome_rows(config):
    path = os.path.join(gu.GOLDEN_DIR, "genome_rows.json")
    if not os.path.exists(path):
        pytest.skip("genome_rows.json not generated (tools/make_golden_full.py genome4 genome5)")
    return json.load(open(path))[f"config{config}"]


def _properties(hot, res, m, cap, plan):
    """What must hold at any size: bins are exact order statistics / exact sums of the compacted depth, the cap holds, the
    moments are those of the array, every implanted event (but the chromosome's last marked run, App. A Q11) is called."""
    rdc = hot.fetch("rd_concat")
    medint = hot.fetch("binmedint")
    sums = hot.fetch("binsum")
    nb = len(medint)
    removed = int(np.sum(res.noncode[1::2] - res.noncode[0::2] + 1))
    assert len(rdc) == plan["n"] - removed and nb == len(rdc) // m
    bins = rdc[:nb * m].reshape(nb, m)
    assert np.array_equal(np.median(bins, axis=1).astype(np.int32), medint)
    assert np.array_equal(bins.sum(axis=1, dtype=np.int64), sums)
    assert rdc.max() <= int(res.stats["cap_median"] * cap)
    x = rdc.astype(np.float64)
    assert res.stats["RDsd"] == pytest.approx(float(np.sqrt((x ** 2).mean() - x.mean() ** 2)), rel=1e-12)
    calls = res.calls("calls")
    missed = []
    for (a, b, code) in plan["events"][:-1]:
        typ = 0 if code in (1, 2) else 1
        if not [c for c in calls if c["type"] == typ and abs(c["start"] - a) <= 2 * m and abs(c["end"] - b) <= 2 * m]:
            missed.append((a, b, code))
    return rdc, missed


def test_config1_60mb_against_reference_golden(hot, hotlib):
    """configs[1]: the 60 Mb Poisson chromosome, -m 101 -NB (the reference's own CPU-runnable size)."""
    from rsicnv_amd import api
    g, plan, flags = _golden("cfg2_60mb")
    plan = _plan(plan)
    assert plan["n"] == 60_000_000
    d_rd, d_fa = _device_case(hotlib, plan)
    res = hot.run_device(api.make_params(**flags), d_rd.data_ptr(), d_fa.data_ptr(), plan["n"])
    _check_against_golden(res, g, "60 Mb")
    rdc, missed = _properties(hot, res, 101, 4.0, plan)
    assert gu.sha(rdc) == str(g["rd_concat_sha"])
    assert len(missed) <= 1, missed


def test_config2_250mb_against_reference_golden(hot, hotlib):
    """configs[2]: the 250 Mb gamma-Poisson chromosome with GC adjustment, -m 101 -NB."""
    from rsicnv_amd import api
    g, plan, flags = _golden("cfg3_250mb")
    plan = _plan(plan)
    d_rd, d_fa = _device_case(hotlib, plan)
    res = hot.run_device(api.make_params(**flags), d_rd.data_ptr(), d_fa.data_ptr(), plan["n"])
    _check_against_golden(res, g, "250 Mb")
    rdc, missed = _properties(hot, res, 101, 4.0, plan)
    assert gu.sha(rdc) == str(g["rd_concat_sha"])          # the whole capped + compacted array, bit for bit
    assert len(missed) <= 2, missed                        # 0.5x / 1.5x events of 3 kb at 30x gamma-Poisson are at the method's limit
    # idempotence: inputs untouched, same calls again
    res2 = hot.run_device(api.make_params(**flags), d_rd.data_ptr(), d_fa.data_ptr(), plan["n"])
    ok, why = calls_equal(res.calls("calls_raw"), res2.calls("calls_raw"), rtol=0)
    assert ok, why
    # the block tests' first round went to the device as one batch (a scan with two dozen segments and more); behind
    # RSI_HOT_BLOCK_BATCH=0 all of them run on the host: the same blocks, the same calls
    hits = dict(hot.phase_times()).get("a15.block batch hits", 0)
    assert hits >= 24, dict(hot.phase_times())
    os.environ["RSI_HOT_BLOCK_BATCH"] = "0"
    try:
        res3 = hot.run_device(api.make_params(**flags), d_rd.data_ptr(), d_fa.data_ptr(), plan["n"])
    finally:
        del os.environ["RSI_HOT_BLOCK_BATCH"]
    assert "a15.block batch hits" not in dict(hot.phase_times())
    for which in ("blocks", "calls_raw", "calls"):
        ok, why = calls_equal(res.calls(which), res3.calls(which))
        assert ok, f"{which}: {why}"


def test_config4_one_60x_med_chromosome_against_reference_golden(hot, hotlib):
    """configs[4]: a 112 Mb chromosome at 60x with -m 51 -MED -cap 4 (small bins, median transform, cap at work)."""
    from rsicnv_amd import api
    g, plan, flags = _golden("cfg5_chr13")
    plan = _plan(plan)
    assert plan["n"] >= 100_000_000 and flags == dict(m=51, trans=1, cap=4.0, gcadjust=1)
    d_rd, d_fa = _device_case(hotlib, plan)
    res = hot.run_device(api.make_params(**flags), d_rd.data_ptr(), d_fa.data_ptr(), plan["n"])
    _check_against_golden(res, g, "60x MED m51")
    rdc, _ = _properties(hot, res, 51, 4.0, plan)
    assert gu.sha(rdc) == str(g["rd_concat_sha"])


@pytest.mark.timeout(900)
def test_config3_genome_in_flight_equals_one_at_a_time_and_golden(hot, hotlib):
    """configs[3]: the 24 chromosomes of the 3 Gb genome, twelve in flight on one GPU: every chromosome's calls and statistics
    equal the same chromosome alone on one context, and ALL 24 equal the golden tables the compiled reference produced from the
    same generated arrays (tests/golden/cfg4_chr1..24.npz: N regions, n', median, SD, raw and final call tables, output rows);
    the genome's rows hash to what bench.py must report (tests/golden/genome_rows.json)."""
    from rsicnv_amd import api, synth
    flags = synth.config_flags(4)
    params = api.make_params(**flags)
    plans = [synth.config_plan(4, chrom=c) for c in range(24)]
    assert abs(sum(p["n"] for p in plans) - 3_000_000_000) < 1000
    bufs = [_device_case(hotlib, p) for p in plans]
    args = [(b[0].data_ptr(), b[1].data_ptr(), p["n"]) for b, p in zip(bufs, plans)]
    pool = api.RsiPool(0, 12)
    batch = pool.run(params, args)
    batch2 = pool.run(params, args)          # a second genome pass through warm workspaces
    ncalls = 0
    for c, (a, b) in enumerate(zip(batch, batch2)):
        one = hot.run_device(params, *args[c])
        for other in (a, b):
            for which in ("calls_raw", "calls"):
                ok, why = calls_equal(one.calls(which), other.calls(which), rtol=0)
                assert ok, f"chromosome {c + 1} {which}: {why}"
            for k in ("RDmedian", "RDsd", "cap_median", "nb_mad", "tmedian1", "tlamda1", "tmedian2", "tlamda2", "Lmax", "n_compact", "nbins"):
                assert one.stats[k] == other.stats[k], (c + 1, k)
        ncalls += len(one.calls("calls"))
        g, gplan, gflags = _golden(f"cfg4_chr{c + 1}")
        assert _plan(gplan)["seed"] == plans[c]["seed"] and _plan(gplan)["n"] == plans[c]["n"] and gflags == flags
        _check_against_golden(a, g, f"chr{c + 1} in flight", chrom=f"chr{c + 1}")
    pool.close()
    ref = _genome_rows(4)
    assert ncalls == ref["calls"] and ncalls >= 300
    assert _genome_hash(batch) == ref["rows_sha256"] == _genome_hash(batch2)


@pytest.mark.timeout(900)
def test_config4_genome_through_the_pool_against_reference_golden(hotlib):
    """configs[4]: the 3 Gb genome at 60x with -m 51 -MED -cap 4, sixteen chromosomes in flight (bench.py's pool): all 24
    chromosomes against the reference's golden tables (tests/golden/cfg5_chr1..24.npz) -- N regions, n', chromosome median and
    SD exactly, raw and final calls, output rows byte for byte -- and the genome's rows hash against genome_rows.json.  Queued
    twice (two genomes in the pool at once, as the bench runs them): both passes must agree."""
    from rsicnv_amd import api, synth
    flags = synth.config_flags(5)
    params = api.make_params(**flags)
    plans = [synth.config_plan(5, chrom=c) for c in range(24)]
    assert abs(sum(p["n"] for p in plans) - 3_000_000_000) < 1000
    bufs = [_device_case(hotlib, p) for p in plans]
    args = [(b[0].data_ptr(), b[1].data_ptr(), p["n"]) for b, p in zip(bufs, plans)]
    pool = api.RsiPool(0, 16)
    h1 = pool.submit(params, args)
    h2 = pool.submit(params, args)
    batch, batch2 = pool.wait(h1), pool.wait(h2)
    for c, r in enumerate(batch):
        g, gplan, gflags = _golden(f"cfg5_chr{c + 1}")
        assert _plan(gplan)["seed"] == plans[c]["seed"] and _plan(gplan)["n"] == plans[c]["n"] and gflags == flags
        _check_against_golden(r, g, f"60x chr{c + 1}", chrom=f"chr{c + 1}")
    ref = _genome_rows(5)
    assert sum(len(r.calls("calls")) for r in batch) == ref["calls"]
    assert _genome_hash(batch) == ref["rows_sha256"] == _genome_hash(batch2)
    pool.close()


def _digest(results):
    """Everything a run reports for its chromosomes, as one comparable tuple (exact: no tolerance anywhere)."""
    out = []
    for r in results:
        st = r.stats
        out.append((tuple(st[k] for k in ("RDmedian", "RDsd", "cap_median", "gc_rdmean", "nb_mad", "nb_r", "nb_tmin", "tmedian1", "tsigma1",
                                          "tlamda1", "tmedian2", "tsigma2", "tlamda2", "Lmax", "n_compact", "nbins", "trim_escapes", "inexact_sums")),
                    tuple(tuple(c[k] for k in ("start", "end", "type", "qscore", "score", "p1", "cnvmed", "cnviqr", "refmed", "refiqr"))
                          for which in ("calls_raw", "calls") for c in r.calls(which))))
    return tuple(out)


@pytest.mark.timeout(240)
def test_soak_of_the_in_kernel_hand_over(hotlib):
    """The kernels hand partial results from workgroup to workgroup without fences (device_util.h: write-through stores, a
    per-wave wait, arrival counters) -- a protocol whose one known failure showed once in a few thousand launches, under
    load.  Soak: a genome of six chromosomes (3 - 13 Mb: hundreds of workgroups per launch, every fold with all its groups)
    through TWO pools at once, several hundred passes each; every pass of either pool must report exactly what the first
    pass reported.  Both pools share the GPU, so launches of one are the uneven load of the other."""
    import threading
    import time
    from rsicnv_amd import api, synth
    plans = [synth.make_plan(n=3_000_017 + 2_000_003 * i, seed=0x50A4 + i, model=1, n_events=6, gaps=2, max_len=30000, end_n=5000,
                             gap_len=8000, centromere=40000) for i in range(6)]
    bufs = [_device_case(hotlib, p) for p in plans]
    args = [(b[0].data_ptr(), b[1].data_ptr(), p["n"]) for b, p in zip(bufs, plans)]
    params = api.make_params()
    pools = [api.RsiPool(0, 6), api.RsiPool(0, 6)]
    first = _digest(pools[0].run(params, args))
    assert sum(len(c[1]) for c in first) > 0
    passes, budget_s = 400, 50.0
    failures, done = [], [0, 0]
    t_end = time.time() + budget_s

    def soak(k):
        for it in range(passes):
            if time.time() > t_end:
                break
            d = _digest(pools[k].run(params, args))
            if d != first:
                bad = [i for i, (x, y) in enumerate(zip(d, first)) if x != y]
                failures.append((k, it, bad))
                break
            done[k] += 1

    threads = [threading.Thread(target=soak, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for p in pools:
        p.close()
    assert not failures, f"pass differs from the first: (pool, pass, chromosomes) {failures}"
    assert min(done) >= 300, f"only {done} passes inside {budget_s} s"
