"""Inputs that push a chromosome off the pipeline's main road (tests/test_fallback_paths.py builds them; the CPU suite checks that
the oracle calls events on them, so that the GPU comparisons are about something)."""
import numpy as np


def wrap_case(n=80_000_000, seed=0x16B17):
    """K2j counts (window GC count, depth byte) pairs in 16-bit LDS fields, a workgroup sees n / 256 bases in strided 1 kb pieces:
    25 Mb without any G or C (GC count 0 everywhere) under a CONSTANT depth put ~98 000 bases of every workgroup into one
    cell -- the fields wrap, the workgroup's sum check raises the flag, the host sends the chromosome through the three-pass
    chain (gccontent.cpp:105-145 computed by K2 + K3' instead).  The other 55 Mb are random sequence under a Poisson depth whose
    mean swings between 26 and 34, so that more than half of the bins differ from the median: with a MAD of zero the reference
    itself aborts (NaN thresholds -> bad_array_new_length in partition_stat_tp), which is no case to test against.  Events in both parts."""
    rng = np.random.default_rng(seed)
    fasta = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].copy()
    lam = 30.0 + 4.0 * np.sin(np.arange(n, dtype=np.float64) * (2 * np.pi / 50_000.0))
    depth = rng.poisson(lam).astype(np.int32)
    del lam
    a0, a1 = 5_000_000, 30_000_000
    fasta[a0:a1] = ord("A")
    depth[a0:a1] = 30
    fasta[:5000] = ord("N"); fasta[-5000:] = ord("N")
    depth[:5000] = 0; depth[-5000:] = 0
    fasta[n // 2:n // 2 + 20000] = ord("N"); depth[n // 2:n // 2 + 20000] = 0
    events = [(8_000_000, 8000, 0.5), (15_000_000, 15000, 1.5), (22_000_000, 5000, 0.0), (45_000_000, 9000, 0.5), (60_000_000, 12000, 1.5),
              (70_000_000, 6000, 0.0)]
    for a, ln, f in events:
        depth[a:a + ln] = (depth[a:a + ln] * f).astype(np.int32)
    return fasta, depth, events


def escape_case(hotlib, n=3_000_011, frac=0.04, seed=0xE5CA):
    """1 - 10 % of the bases at 255x and more: too many for K2j's workgroups to list (63 each), far fewer than the eighth of the
    chromosome that sends it down the int32 path -- the middle regime, where k_escape_hist adds the rescaled escapes to the value
    histogram before the cap median is walked (loaddata.cpp:229-240 on gccontent.cpp:89's values)."""
    from conftest import make_case
    plan, fasta, depth = make_case(hotlib, dict(n=n, seed=seed, model=1, n_events=8, gaps=1, max_len=30000, end_n=4000, gap_len=9000))
    rng = np.random.default_rng(seed)
    depth = depth.copy()
    # amplified stretches (a 12x amplicon every so often) plus a sprinkle of single bases
    nstretch = 12
    ln = int(n * frac * 0.8 / nstretch)
    for s in rng.integers(10_000, n - ln - 10_000, size=nstretch):
        depth[s:s + ln] = depth[s:s + ln] * 12 + rng.integers(0, 12, size=ln)
    idx = rng.integers(10_000, n - 10_000, size=int(n * frac * 0.2))
    depth[idx] = 255 + rng.integers(0, 2000, size=idx.size)
    depth[fasta == ord("N")] = 0
    return plan, fasta, depth.astype(np.int32)
