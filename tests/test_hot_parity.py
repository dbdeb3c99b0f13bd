"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Integer / index work is bit-exact; NB floats and SCORE to 1e-6 relative (north_star tolerance)."""
import numpy as np
import pytest

from conftest import calls_equal, make_case, small_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hot():
    from rsicnv_amd import api
    h = api.RsiHot(0)
    yield h
    h.close()


@pytest.mark.parametrize("name,plan_kw,flag_kw", small_cases(), ids=[c[0] for c in small_cases()])
def test_stages_and_calls(hot, hotlib, oracle_cls, name, plan_kw, flag_kw):
    import oracle
    from rsicnv_amd import api
    plan, fasta, depth = make_case(hotlib, plan_kw)
    O = oracle_cls()
    O.run(oracle.make_params(**flag_kw), depth, fasta)
    res = hot.run(api.make_params(**flag_kw), depth, fasta)
    st = res.stats
    # A1: padded N regions
    assert np.array_equal(res.noncode, O.i32("noncode"))
    # A2/A3: GC-corrected depth
    if flag_kw.get("gcadjust", 1):
        assert np.array_equal(hot.fetch("rd_gc"), O.i32("rd_gc"))
        assert st["gc_rdmean"] == O.f64("chrom")[3]
    # A4/A5: capped + compacted depth, A6: chromosome median / SD
    assert np.array_equal(hot.fetch("rd_concat"), O.i32("rd_concat"))
    assert st["RDmedian"] == O.f64("chrom")[0]
    assert st["RDsd"] == O.f64("chrom")[1]
    if flag_kw.get("cap", 4.0) > 1:
        assert st["cap_median"] == O.f64("chrom")[2]
    # A8: per-bin medians
    assert np.array_equal(hot.fetch("binmedint"), O.i32("binmedint"))
    # A9/A10: MAD, r, NB transform
    nbs = O.f64("nb")
    assert st["nb_mad"] == nbs[1] and st["nb_r"] == nbs[2]
    np.testing.assert_allclose(hot.fetch("binnb"), O.f32("binnb"), rtol=1e-6, atol=0)
    trans = flag_kw.get("trans", 0)
    key = "scan_med" if trans == 1 else "scan_nb"
    pre = "med" if trans == 1 else "nb"
    sc = O.f64(key)
    got = [st[k] for k in ("tmedian1", "tsigma1", "tlamda1", "tmedian2", "tsigma2", "tlamda2")]
    np.testing.assert_allclose(got, sc[:6], rtol=1e-12)
    assert st["Lmax"] == int(sc[7])
    # trim walks that left the array (marked nothing, counted: DESIGN divergence 1); under -ALL both scans run and the library
    # reports their sum
    exp_escapes = int(O.f64("scan_med")[9]) + int(O.f64("scan_nb")[9]) if trans == 2 else int(sc[9])
    assert st["trim_escapes"] == exp_escapes
    assert st["inexact_sums"] == 0
    # A12/A13: status arrays of the (last) scan
    assert np.array_equal(hot.fetch("status1"), O.i32(f"{pre}_status1"))
    assert np.array_equal(hot.fetch("status1f"), O.i32(f"{pre}_status1f"))
    assert np.array_equal(hot.fetch("status2"), O.i32(f"{pre}_status2"))
    # A14: segments; A15..A19: calls
    segs_o = (O.calls("segs_med") if trans != 0 else []) + (O.calls("segs_nb") if trans != 1 else [])
    for which, exp in (("segs", segs_o), ("blocks", O.calls("blocks")), ("calls_raw", O.calls("calls_raw")), ("calls", O.calls("calls"))):
        ok, why = calls_equal(res.calls(which), exp)
        assert ok, f"{name} {which}: {why}"
    assert len(res.calls("calls_raw")) >= 1, "the case should call at least one implanted event"
