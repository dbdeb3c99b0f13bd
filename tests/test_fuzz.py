"""GPU: random chromosomes (length, depth model and mean, events, gaps, pile-ups, uncovered stretches, assembly gaps) under random
flags (-m 11 ... 439, -NB / -MED / -ALL, caps, -NOGC, -nomerge) through the library and through the oracle in a child process:
every array, scalar, status vector and call list must agree, and where the oracle refuses (the reference exits or aborts) the
library must refuse too.  tools/fuzz_probe.py is the same loop for longer sessions (40 cases of seed 1 ran clean in round 4)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", [11, 12])
def test_random_chromosomes_and_flags_agree_with_the_oracle(seed):
    import fuzz_probe
    assert fuzz_probe.run(8, seed, max_bins=60_000) == 0


def test_random_deep_chromosomes_agree_with_the_oracle():
    """120x / 300x under caps of 2, 4 and 8 times the median, bins of 11 ... 439 bases: the 16-bit compaction kernel (K4w) in
    every thread-per-bin class, the quantile histograms' sampled LDS window, the int32 candidate kernels."""
    import fuzz_probe
    assert fuzz_probe.run(8, 13, max_bins=60_000, deep=True) == 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("flags", [dict(), dict(m=51, trans=1), dict(gcadjust=0, cap=2.0), dict(m=201, trans=2, cap=-1.0)])
def test_a_pool_of_random_chromosomes_equals_one_context(flags):
    """Ten random chromosomes of very different lengths (40 kb ... 3 Mb; one of them deep, one with a pile-up, one without spread enough to
    call anything) through a pool of eight workers from host memory, twice, against the same chromosomes one at a time on a
    single context: the same statistics and lists, bit for bit (the single-context results are what the oracle tests pin)."""
    import numpy as np
    from conftest import make_case
    from rsicnv_amd import api
    lib = api.load_library()
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
    params = api.make_params(**flags)
    hot = api.RsiHot(0)
    single = []
    for fasta, depth in cases:
        try:
            single.append(hot.run(params, depth, fasta))
        except api.RsiError as e:
            single.append(e)
    usable = [i for i, r in enumerate(single) if not isinstance(r, api.RsiError)]
    assert len(usable) >= 6
    chroms = [(cases[i][1].ctypes.data, cases[i][0].ctypes.data, cases[i][1].size) for i in usable]
    pool = api.RsiPool(0, 8)
    keys = ("RDmedian", "RDsd", "cap_median", "nb_mad", "nb_r", "tmedian1", "tlamda1", "tmedian2", "tlamda2", "Lmax", "n_compact", "nbins", "trim_escapes")
    for _ in range(2):
        batch = pool.run(params, chroms, host=True)
        for i, got in zip(usable, batch):
            want = single[i]
            assert [got.stats[k] for k in keys] == [want.stats[k] for k in keys], i
            for which in ("blocks", "calls_raw", "calls"):
                assert got.calls(which) == want.calls(which), (i, which)
            assert np.array_equal(got.noncode, want.noncode)
    pool.close()
    hot.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("flags_index", [1, 2])
def test_results_do_not_depend_on_what_fresh_allocations_hold(flags_index):
    """RSI_HOT_POISON=1 fills every device and pinned allocation with 0xA5 (instead of the zero pages a fresh process gets, or another
    context's leftovers in a long one).  Ten chromosomes of different sizes through ONE context, so that its buffers grow on the
    way, each compared with the oracle (tools/uninit_probe.py, a child process: the switch is read once per process).  Round 4
    found the split candidate tests starting from stale counters this way: a buffer that had grown at the same address was taken
    for one that had not been reallocated."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSI_HOT_POISON="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "uninit_probe.py"), str(flags_index), "1"], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "ALL SAME" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
