"""k_fs_* (filterstatus' level sums) alone on synthetic bins: how the time depends on the number of bins, the share of marked
bins and the number of levels they spread over.  usage: python tools/fs_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rsicnv_amd import api
hot = api.RsiHot(0)
hot.set_timing(1)
rng = np.random.default_rng(7)
Lmax = 99
for nb in (590_000, 2_450_000):
    T = (60.0 + 4.0 * rng.standard_normal(nb)).astype(np.float32).clip(0.5, None)
    for frac, nlev in ((0.0, 1), (0.02, 10), (0.02, 150), (0.15, 10), (0.15, 150)):
        st = np.zeros(nb, dtype=np.int32)
        nrun = int(nb * frac / 200)
        levels = rng.integers(1, Lmax + 1, size=max(nlev, 1)) * rng.choice([-1, 1], size=max(nlev, 1))
        for s0 in rng.integers(0, nb - 400, size=nrun):
            st[s0:s0 + 200] = levels[rng.integers(0, len(levels))]
        ts = []
        for it in range(5):
            hot.debug_level_sums(T, st, Lmax)
            ts.append(dict(hot.kernel_times()).get("level_sums", 0.0))
        print(f"nb {nb} marked {np.count_nonzero(st)/nb:.3f} levels {len(np.unique(st))-1}: level_sums (4 launches) {min(ts[1:])*1e3:.0f} us", flush=True)
hot.close()
