import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hotlib():
    from rsicnv_amd import api
    return api.load_library()


@pytest.fixture(scope="session")
def oracle_cls():
    import oracle
    return oracle.Oracle


def small_cases():
    """(name, plan kwargs, flag kwargs): the parity matrix shared by the CPU and GPU tests.
    Lengths are chosen to hit the n mod 20 tail quirks as well."""
    return [
        ("poisson_nb_m101", dict(n=400_000, seed=0xA11CE, model=0, n_events=5, gaps=1, max_len=20000, end_n=5000, gap_len=8000), dict()),
        ("poisson_tail7", dict(n=400_007, seed=0xA11CF, model=0, n_events=5, gaps=1, max_len=20000, end_n=5000, gap_len=8000), dict()),
        ("poisson_tail1", dict(n=300_001, seed=0xA11D0, model=0, n_events=4, gaps=0, max_len=15000, end_n=4000), dict()),
        ("gampois_nb_m101", dict(n=600_000, seed=0xB0B, model=1, n_events=6, gaps=2, max_len=30000, end_n=5000, gap_len=6000, centromere=20000), dict()),
        ("gampois_med_m51_cap4", dict(n=500_013, seed=0xB0C, model=1, mean=60.0, n_events=6, gaps=1, max_len=20000, end_n=5000, gap_len=6000), dict(m=51, trans=1, cap=4.0)),
        ("poisson_nogc", dict(n=400_000, seed=0xA11D1, model=0, n_events=5, gaps=1, max_len=20000, end_n=5000, gap_len=8000), dict(gcadjust=0)),
        ("poisson_nocap", dict(n=350_019, seed=0xA11D2, model=0, n_events=4, gaps=1, max_len=20000, end_n=5000, gap_len=8000), dict(cap=-1.0)),
        ("gampois_all", dict(n=450_000, seed=0xB0D, model=1, n_events=5, gaps=1, max_len=20000, end_n=5000, gap_len=6000), dict(trans=2)),
        ("poisson_no_n", dict(n=300_000, seed=0xA11D3, model=0, n_events=4, gaps=0, max_len=15000, end_n=0), dict()),
        # no N at the chromosome ends: the clamped GC windows (Q1) and the tail cells of the 20-slice write-back (Q2/Q3)
        # survive into the compacted array, where the kernel that rescales from the byte copy has to reproduce them
        ("poisson_no_n_tail13", dict(n=300_013, seed=0xA11D4, model=0, n_events=4, gaps=0, max_len=15000, end_n=0), dict()),
        ("gampois_no_n_tail1", dict(n=320_001, seed=0xB0E, model=1, n_events=4, gaps=1, max_len=15000, end_n=0, gap_len=5000), dict(m=51, trans=1)),
        ("gampois_no_n_tail19", dict(n=280_019, seed=0xB0F, model=1, mean=45.0, n_events=4, gaps=0, max_len=15000, end_n=0), dict()),
        # wide bins through the byte kernels: 32 bins per tile with eight threads per bin (m <= 216), 16 with sixteen (m <= 440)
        ("gampois_nb_m201", dict(n=700_000, seed=0xB10, model=1, n_events=5, gaps=1, max_len=40000, end_n=5000, gap_len=8000), dict(m=201)),
        ("poisson_med_m301_tail5", dict(n=900_005, seed=0xA11D5, model=0, mean=50.0, n_events=5, gaps=1, max_len=60000, end_n=6000, gap_len=9000), dict(m=301, trans=1)),
        ("gampois_nb_m439_cap3", dict(n=1_200_003, seed=0xB11, model=1, mean=35.0, n_events=5, gaps=1, max_len=80000, end_n=6000, gap_len=9000), dict(m=439, cap=3.0)),
    ]


def wide_scan_cases():
    """Small bins make the scan long (Lmax = 10000/m, rsi.cpp:1830): 909, 1428 and 2000 window lengths
    per bin, and 3333 at -m 3.  They exercise the scan kernel's big tiles, its capped mark levels and the
    chromosome-end tiles.  Kept short: the reference needs minutes for them (golden files only)."""
    return [
        ("wide_m11_nb", dict(n=80_007, seed=0x5CA1, model=0, n_events=3, gaps=1, max_len=4000, end_n=1500, gap_len=2000), dict(m=11)),
        ("wide_m7_med", dict(n=60_003, seed=0x5CA2, model=1, mean=45.0, n_events=3, gaps=0, max_len=3000, end_n=1000), dict(m=7, trans=1)),
        ("wide_m5_nb", dict(n=60_001, seed=0x5CA3, model=0, mean=40.0, n_events=3, gaps=0, max_len=2500, end_n=1000), dict(m=5)),
        # -m 3: Lmax = 3333, beyond the 2048 the scan kernel used to stop at (one 256-bin tile + 2 x 1667 halo bins in LDS)
        ("wide_m3_nb", dict(n=45_003, seed=0x5CA4, model=0, mean=40.0, n_events=3, gaps=0, max_len=2000, end_n=900), dict(m=3)),
    ]


def make_case(lib, plan_kw):
    from rsicnv_amd import synth
    plan = synth.make_plan(**plan_kw)
    fasta, depth = synth.generate_host(lib, plan)
    return plan, fasta, depth


def calls_equal(a, b, rtol=1e-6):
    """START/END/TYPE (and the integer fields) exact; SCORE and the float statistics to rtol."""
    if len(a) != len(b):
        return False, f"count {len(a)} != {len(b)}"
    for i, (x, y) in enumerate(zip(a, b)):
        for k in ("start", "end", "type", "geno", "status", "length", "qscore"):
            if x[k] != y[k]:
                return False, f"call {i} field {k}: {x[k]} != {y[k]}"
        for k in ("score", "p1", "cnvmed", "cnvsd", "cnviqr", "refmed", "refsd", "refiqr"):
            if not np.isclose(x[k], y[k], rtol=rtol, atol=1e-300):
                return False, f"call {i} field {k}: {x[k]} != {y[k]}"
    return True, ""
