"""CPU: the oracle (oracle/rsi_oracle.cpp) against the golden outputs of the real reference
(tests/golden/*.npz, tools/make_golden.py).  This is what pins the oracle on machines without
/root/reference.  Integer arrays bit-exact; floats bit-exact too here (same libm, same rounding)."""
import numpy as np
import pytest

import golden_util as gu
from conftest import calls_equal, small_cases, wide_scan_cases

# one long-scan case pins the oracle there too (they take ~20 s of oracle time each; the GPU suite checks all three
# against the same golden files)
NAMES = [c[0] for c in small_cases()] + [wide_scan_cases()[0][0]]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_golden(hotlib, oracle_cls, name):
    import oracle
    g, plan, flags = gu.load(name)
    fasta, depth = gu.regenerate_inputs(hotlib, plan, g)
    O = oracle_cls()
    O.run(oracle.make_params(**flags), depth, fasta)
    assert np.array_equal(O.i32("noncode"), g["noncode"])
    if flags.get("gcadjust", 1):
        assert gu.sha(O.i32("rd_gc")) == str(g["rd_gc_sha"])
    assert gu.sha(O.i32("rd_cap")) == str(g["rd_cap_sha"])
    assert gu.sha(O.i32("rd_concat")) == str(g["rd_concat_sha"])
    if "rd_gc" in g.files:
        assert np.array_equal(O.i32("rd_gc"), g["rd_gc"]) and np.array_equal(O.i32("rd_concat"), g["rd_concat"])
    assert tuple(O.f64("chrom")[:2]) == tuple(g["chrom"])
    assert np.array_equal(O.i32("binmedint"), g["binmedint"])
    assert np.array_equal(O.f32("binmed"), g["binmed"])
    assert np.array_equal(O.f32("binnb"), g["binnb"])
    trans = flags.get("trans", 0)
    for pre in (["med"] if trans == 1 else ["nb"] if trans == 0 else ["med", "nb"]):
        sc = O.f64(f"scan_{pre}")
        assert np.array_equal(sc[:9], g[f"{pre}_scan"])
        for w in ("status1", "status1f", "status2"):
            assert np.array_equal(O.i32(f"{pre}_{w}"), g[f"{pre}_{w}"]), f"{pre}_{w}"
        ok, why = calls_equal(O.calls(f"segs_{pre}"), gu.calls_from_array(g[f"{pre}_segs"]), rtol=0)
        assert ok, why
    for which in ("calls_raw", "calls"):
        ok, why = calls_equal(O.calls(which), gu.calls_from_array(g[which]), rtol=0)
        assert ok, f"{which}: {why}"


def test_rsistatus_restatement_equals_the_oracle(hotlib, oracle_cls):
    """CPU: rsistatus_numpy (tests/scan_restatement.py, the checker of the -m 1 scan in test_hot_extra.py) against the oracle's own rsistatus where the oracle is fast
    enough (-m 11, Lmax 909; the oracle itself is pinned by the golden file of the same case)."""
    import oracle
    from conftest import make_case
    from scan_restatement import rsistatus_numpy
    name, plan_kw, flag_kw = wide_scan_cases()[0]
    _, fasta, depth = make_case(hotlib, plan_kw)
    O = oracle_cls()
    O.run(oracle.make_params(**flag_kw), depth, fasta)
    sc = O.f64("scan_nb")
    exp = rsistatus_numpy(O.f32("binnb"), O.i32("binmedint"), O.f64("chrom")[0], sc[0], sc[2], int(sc[7]), O.exact_median)
    assert np.count_nonzero(exp) > 100
    assert np.array_equal(exp, O.i32("nb_status1"))
