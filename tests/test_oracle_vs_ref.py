"""CPU, only where oracle/_ref/libref.so exists (this container; it also travels to the GPU box):
the oracle's numeric utilities and a larger whole-path case against the compiled reference itself."""
import numpy as np
import pytest

import oracle
from conftest import calls_equal, make_case

pytestmark = pytest.mark.skipif(not oracle.ref_available(), reason="oracle/_ref/libref.so not built (needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    return oracle.Ref()


def test_histogram_quantiles_match(ref, oracle_cls):
    O = oracle_cls()
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 4, 5, 31, 100, 1001, 50_000):
        xi = rng.poisson(30, n).astype(np.int32)
        xf = (rng.gamma(9.0, 3.3, n)).astype(np.float32)
        xd = xf.astype(np.float64) * 1.37
        for x in (xi, xf, xd):
            assert O.median(x) == ref.median(x)
        assert O.iqr(xi) == ref.iqr(xi) and O.iqr(xf) == ref.iqr(xf)
    # degenerate: all equal / within the grid step -> the mean is returned
    c = np.full(17, 5, dtype=np.int32)
    assert O.median(c) == ref.median(c) == 5.0
    f = (30.0 + 0.001 * np.arange(9)).astype(np.float32)
    assert O.median(f) == ref.median(f)


def test_exact_median_and_pnorm_match(ref, oracle_cls):
    O = oracle_cls()
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 4, 7, 8, 99, 100, 196):
        x = rng.integers(0, 60, n).astype(np.int32)
        assert O.exact_median(x) == ref.exact_median(x) == float(np.median(x))
    for v in np.concatenate([np.linspace(-16, 16, 1281), rng.normal(0, 4, 500)]):
        assert O.pnorm(float(v)) == ref.pnorm(float(v))


@pytest.mark.parametrize("flags", [dict(), dict(m=51, trans=1), dict(gcadjust=0, cap=-1.0), dict(trans=2),
                                   dict(epsilon=2.5, chklen=1.5, merge=0), dict(trans=1, threshold=0.8, maxchkbp=2000, m=75)],
                         ids=["nb", "med51", "nogc_nocap", "all", "eps_reflen_nomerge", "med_threshold_maxchkbp_m75"])
def test_whole_path_2mb(ref, oracle_cls, hotlib, flags):
    plan_kw = dict(n=2_000_003, seed=0xD00D + len(flags), model=1, n_events=9, gaps=2, max_len=60000, end_n=10000, gap_len=30000)
    _, fasta, depth = make_case(hotlib, plan_kw)
    p = oracle.make_params(**flags)
    O = oracle_cls()
    O.run(p, depth, fasta)
    ref.load(p, depth, fasta)
    assert np.array_equal(ref.noncode(), O.i32("noncode"))
    ref.stage_gc()
    if flags.get("gcadjust", 1):
        assert np.array_equal(ref.rd(), O.i32("rd_gc"))
    ref.stage_cap()
    assert np.array_equal(ref.rd(), O.i32("rd_cap"))
    ref.stage_concat()
    assert np.array_equal(ref.rd(), O.i32("rd_concat"))
    assert ref.chrom_scalars() == tuple(O.f64("chrom")[:2])
    med, medint, nbn = ref.stage_bins()
    assert np.array_equal(medint, O.i32("binmedint")) and np.array_equal(nbn, O.f32("binnb")) and np.array_equal(med, O.f32("binmed"))
    trans = flags.get("trans", 0)
    for use_med in ([True] if trans == 1 else [False] if trans == 0 else [True, False]):
        sc, st, segs = ref.scan(use_med)
        assert sc.stepwise_matches_reference == 1
        pre = "med" if use_med else "nb"
        so = O.f64(f"scan_{pre}")
        assert (sc.tmedian1, sc.tsigma1, sc.tlamda1, sc.tmedian2, sc.tsigma2, sc.tlamda2) == tuple(so[:6])
        assert sc.Lmax == int(so[7])
        for i, w in enumerate(("status1", "status1f", "status2")):
            assert np.array_equal(st[i], O.i32(f"{pre}_{w}"))
        ok, why = calls_equal(O.calls(f"segs_{pre}"), segs, rtol=0)
        assert ok, why
    raw, fin, _ = ref.detect()
    for which, exp in (("calls_raw", raw), ("calls", fin)):
        ok, why = calls_equal(O.calls(which), exp, rtol=0)
        assert ok, f"{which}: {why}"
    assert len(raw) >= 3

