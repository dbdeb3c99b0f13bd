"""Test infrastructure: a vectorised restatement of the reference's scan, for window lengths the oracle cannot reach in test time."""
import numpy as np


def rsistatus_numpy(T, medint, RDmedian, tmedian, tlamda, Lmax, exact_median):
    """rsistatus (rsi.cpp:1191-1259) with runmeantp (wufunctions.cpp:573-647), restated so that it finishes in seconds at
    Lmax = 10000: the window sums of ALL lengths advance together, one start position per step (for every length the very
    sequence of double operations of the reference's loop: sum = sum - y[first-1] + y[last]), the score test is vectorised, and
    the (few) hits are replayed in the reference's order -- deletions before duplications, lengths ascending, positions
    ascending; median test, the four trimming walks, first mark wins, stop at 20 % marked."""
    nb = T.size
    y = T.astype(np.float64)
    Ls = np.arange(1, Lmax + 1)
    h = Ls // 2
    sums = np.cumsum(y)[Ls - 1].copy()            # sequential double sums of y[0 .. L-1]
    rootL = np.sqrt(Ls.astype(np.float64))
    hits = {0: [], 1: []}
    for first in range(1, nb):
        last = first + Ls - 1
        live = last < nb
        if not live.any():
            break
        sums[live] = sums[live] - y[first - 1] + y[last[live]]
        smo = (sums / Ls).astype(np.float32).astype(np.float64)
        i = h + first                              # the position this window's mean belongs to
        scan = live & (i >= h + 1) & (i < nb - h - 1)
        score = (smo - tmedian) * rootL
        for sweep, cond in ((0, score <= -tlamda), (1, score >= tlamda)):
            for k in np.nonzero(scan & cond)[0]:
                hits[sweep].append((int(Ls[k]), int(i[k])))
    st = np.zeros(nb, dtype=np.int32)
    for sweep in (0, 1):
        dele = sweep == 0
        lim = RDmedian * 0.75 if dele else RDmedian * 1.25
        by_len = {}
        for L, pos in hits[sweep]:
            by_len.setdefault(L, []).append(pos)
        for L in sorted(by_len):
            for pos in sorted(by_len[L]):
                i1 = pos - L // 2
                i2 = i1 + L - 1
                wm = exact_median(medint[i1:i2 + 1])
                if (wm > lim) if dele else (wm < lim):
                    continue
                if dele:
                    while T[i1] > tmedian: i1 += 1
                    while medint[i1] > lim: i1 += 1
                    while T[i2] > tmedian: i2 -= 1
                    while medint[i2] > lim: i2 -= 1
                else:
                    while T[i1] < tmedian: i1 += 1
                    while medint[i1] < lim: i1 += 1
                    while T[i2] < tmedian: i2 -= 1
                    while medint[i2] < lim: i2 -= 1
                seg = st[i1:i2 + 1]
                seg[seg == 0] = -L if dele else L
            marked = np.count_nonzero(st < 0) if dele else np.count_nonzero(st > 0)
            if marked / nb > 0.2:
                break
    return st


def long_scan_case(nb=36_000, Lmax=10_000, plateau=(18_000, 23_000), quiet=(11_000, 30_000)):
    """Bins for one scan pass at Lmax = 10000: noise of one sigma at both ends with a short deletion and a short duplication
    (hits at 6 .. 15 bins), a quiet stretch in between with a shallow plateau of 5000 bins that only windows of about its own
    length detect (hits at 4980 .. 5020 bins: the sweep's longest prefixes, a few hundred hits instead of the millions a
    deep event makes at this bin size).  Returns T, medint, RDmedian, tmedian, tlamda, Lmax.
    The arguments stretch the same picture for the scans beyond 10 400 lengths (a computed length, rsi.cpp:1286-1289)."""
    rng = np.random.default_rng(0x5CA5)
    T = np.full(nb, 40.0, dtype=np.float32)
    noisy = np.r_[0:quiet[0], quiet[1]:nb]
    T[noisy] += rng.standard_normal(noisy.size).astype(np.float32)
    medint = np.full(nb, 40, dtype=np.int32)
    T[3000:3040] -= 3.0
    medint[3000:3040] = 20
    T[7000:7050] += 3.0
    medint[7000:7050] = 60
    width = plateau[1] - plateau[0]
    T[plateau[0]:plateau[1]] = np.float32(40.0) - np.float32(10.45 / np.sqrt(width - 20.0))
    medint[plateau[0]:plateau[1]] = 29
    return T, medint, 40.0, 40.0, 10.45, Lmax
