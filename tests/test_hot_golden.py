"""GPU: the HIP path against the golden outputs of the real reference (tests/golden/*.npz)."""
import numpy as np
import pytest

import golden_util as gu
from conftest import calls_equal, small_cases, wide_scan_cases

pytestmark = pytest.mark.gpu
NAMES = [c[0] for c in small_cases()] + [c[0] for c in wide_scan_cases()]


@pytest.fixture(scope="module")
def hot():
    from rsicnv_amd import api
    h = api.RsiHot(0)
    yield h
    h.close()


@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_reference_golden(hot, hotlib, name):
    from rsicnv_amd import api
    g, plan, flags = gu.load(name)
    fasta, depth = gu.regenerate_inputs(hotlib, plan, g)
    res = hot.run(api.make_params(**flags), depth, fasta)
    assert np.array_equal(res.noncode, g["noncode"])
    if flags.get("gcadjust", 1):
        assert gu.sha(hot.fetch("rd_gc")) == str(g["rd_gc_sha"])
    assert gu.sha(hot.fetch("rd_concat")) == str(g["rd_concat_sha"])
    assert (res.stats["RDmedian"], res.stats["RDsd"]) == tuple(g["chrom"])
    assert np.array_equal(hot.fetch("binmedint"), g["binmedint"])
    np.testing.assert_allclose(hot.fetch("binnb"), g["binnb"], rtol=1e-6, atol=0)
    trans = flags.get("trans", 0)
    pre = "med" if trans == 1 else "nb"      # the last scan run leaves its arrays on the device
    sc = g[f"{pre}_scan"]
    got = [res.stats[k] for k in ("tmedian1", "tsigma1", "tlamda1", "tmedian2", "tsigma2", "tlamda2")]
    np.testing.assert_allclose(got, sc[:6], rtol=1e-12)
    assert res.stats["Lmax"] == int(sc[7])
    for w in ("status1", "status1f", "status2"):
        assert np.array_equal(hot.fetch(w), g[f"{pre}_{w}"]), w
    for which in ("calls_raw", "calls"):
        ok, why = calls_equal(res.calls(which), gu.calls_from_array(g[which]))
        assert ok, f"{which}: {why}"
    # the output rows, byte for byte as cnv_format1 prints them (rsi.cpp:581-631)
    assert "\n".join(res.format_rows("chrS")) + ("\n" if res.rows else "") == str(g["rows"])
