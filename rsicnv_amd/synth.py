"""Synthetic chromosome plans for the BASELINE.json configurations.

A *plan* is a dict(seed, n, model, mean, nb_size, events, nruns, lower) that the generators in
librsi_hot.so (include/rsi_synth.h) turn into FASTA bytes + per-base depth, on the host or on the
GPU, bit-identically.  Event placement is decided here with a tiny integer PRNG so that the same
plan is produced everywhere.
"""
import ctypes as C

import numpy as np

CN_1X, CN_0X, CN_HALF, CN_1P5, CN_2X = 0, 1, 2, 3, 4
_MASK = (1 << 64) - 1

# GRCh37-like chromosome lengths in Mb (chr1-22, X, Y), SURVEY.md section 8d config 4
GENOME_MB = [249, 243, 198, 191, 181, 171, 159, 146, 141, 136, 135, 133, 115, 107, 102, 90, 81, 78, 59, 63, 48, 51, 155, 59]


class Interval(C.Structure):
    _fields_ = [("beg", C.c_int64), ("end", C.c_int64), ("code", C.c_int32), ("pad", C.c_int32)]


class Spec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n", C.c_int64), ("model", C.c_int32), ("n_events", C.c_int32),
                ("n_nruns", C.c_int32), ("n_lower", C.c_int32), ("mean", C.c_double), ("nb_size", C.c_double),
                ("events", C.POINTER(Interval)), ("nruns", C.POINTER(Interval)), ("lower", C.POINTER(Interval))]


def _mix(z):
    z &= _MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


class _Rng:
    def __init__(self, seed):
        self.s = seed & _MASK

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _MASK
        return _mix(self.s)

    def below(self, k):
        return self.next() % k


def make_plan(n, seed, model=0, mean=30.0, nb_size=10.0, n_events=9, gaps=1, min_len=3000, max_len=100000,
              end_n=10000, gap_len=50000, centromere=0):
    """N runs: `end_n` bases at each end, `gaps` internal gaps of `gap_len`, optional centromere.
    Events: the DEL/DUP mix of SURVEY.md 8d (0.5x, 0x, 1.5x, 2x), lengths in [min_len, max_len],
    placed in disjoint slots away from the N runs.  One soft-masked (lower-case) stretch."""
    rng = _Rng(seed ^ 0x504C414E)
    nruns = []
    if end_n > 0 and n > 4 * end_n:
        nruns.append((0, end_n))
        nruns.append((n - end_n, n))
    slots = max(n_events + gaps + 2, 4)
    slot = n // slots
    order = list(range(1, slots - 1))
    # place gaps/centromere in the middle slots, events in the others
    taken = set()
    mid = slots // 2
    if centromere > 0 and centromere < slot:
        a = mid * slot + (slot - centromere) // 2
        nruns.append((a, a + centromere))
        taken.add(mid)
    g = 0
    for s in order:
        if g >= gaps:
            break
        if s in taken or s % 2 == 0:
            continue
        if gap_len < slot // 2:
            a = s * slot + slot // 4 + rng.below(max(slot // 4, 1))
            nruns.append((a, a + gap_len))
            taken.add(s)
            g += 1
    codes = [CN_HALF, CN_1P5, CN_HALF, CN_0X, CN_1P5, CN_HALF, CN_2X, CN_1P5, CN_HALF]
    events = []
    e = 0
    for s in order:
        if e >= n_events:
            break
        if s in taken:
            continue
        hi = min(max_len, slot // 3)
        lo = min(min_len, hi)
        ln = lo + rng.below(hi - lo + 1)
        a = s * slot + slot // 6 + rng.below(max(slot // 3, 1))
        events.append((a, a + ln, codes[e % len(codes)]))
        e += 1
    nruns.sort()
    events.sort()
    # one soft-masked stretch in the first free slot, not overlapping an event
    lower = []
    for s in order:
        if s not in taken and all(not (ev[0] < (s + 1) * slot and ev[1] > s * slot + slot * 5 // 6) for ev in events):
            a = s * slot + slot * 5 // 6
            lower.append((a, min(a + max(2000, slot // 50), (s + 1) * slot)))
            break
    return dict(seed=seed, n=int(n), model=model, mean=float(mean), nb_size=float(nb_size), events=events,
                nruns=nruns, lower=lower)


def config_plan(config, chrom=0, scale=1.0):
    """Plans for BASELINE.json configs 2-5 (1-based index as in SURVEY.md 8d).  `scale` shrinks the
    chromosome lengths (tests); the layout rules stay the same."""
    if config == 2:
        return make_plan(int(60_000_000 * scale), 0x5EED0002, model=0, mean=30.0, n_events=9, gaps=1)
    if config == 3:
        return make_plan(int(250_000_000 * scale), 0x5EED0003, model=1, mean=30.0, n_events=40, gaps=5,
                         centromere=int(3_000_000 * scale))
    if config in (4, 5):
        tot = sum(GENOME_MB)
        n = int(GENOME_MB[chrom] * 1_000_000 * 3000 / tot * scale)
        seed = (0x5EED0004 if config == 4 else 0x5EED0005) + chrom
        return make_plan(n, seed, model=1, mean=30.0 if config == 4 else 60.0, n_events=20, gaps=2,
                         centromere=int(1_000_000 * scale))
    raise ValueError(config)


def config_flags(config):
    """Reference flags for a config: dict(m, trans, cap, gcadjust)."""
    if config == 5:
        return dict(m=51, trans=1, cap=4.0, gcadjust=1)
    return dict(m=101, trans=0, cap=4.0, gcadjust=1)


def _intervals(lst):
    arr = (Interval * max(len(lst), 1))()
    for i, t in enumerate(lst):
        arr[i].beg, arr[i].end = t[0], t[1]
        arr[i].code = t[2] if len(t) > 2 else 0
    return arr


def make_spec(plan):
    """ctypes Spec plus the arrays it points to (keep the tuple alive while the spec is used)."""
    ev, nr, lo = _intervals(plan["events"]), _intervals(plan["nruns"]), _intervals(plan["lower"])
    spec = Spec(plan["seed"], plan["n"], plan["model"], len(plan["events"]), len(plan["nruns"]), len(plan["lower"]),
                plan["mean"], plan["nb_size"], ev, nr, lo)
    return spec, (ev, nr, lo)


def generate_host(lib, plan):
    """FASTA (uint8[n]) and depth (int32[n]) on the host through rsi_synth_generate_host."""
    spec, keep = make_spec(plan)
    n = plan["n"]
    fasta = np.empty(n, dtype=np.uint8)
    depth = np.empty(n, dtype=np.int32)
    lib.rsi_synth_generate_host.argtypes = [C.POINTER(Spec), C.c_void_p, C.c_void_p]
    rc = lib.rsi_synth_generate_host(C.byref(spec), fasta.ctypes.data, depth.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"rsi_synth_generate_host failed: {rc}")
    del keep
    return fasta, depth


def generate_device(lib, plan, d_fasta_ptr, d_depth_ptr, stream=0):
    """Fill device buffers (raw pointers, e.g. torch tensor .data_ptr()) on the GPU."""
    spec, keep = make_spec(plan)
    lib.rsi_synth_generate_device.argtypes = [C.POINTER(Spec), C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.rsi_synth_generate_device(C.byref(spec), C.c_void_p(d_fasta_ptr), C.c_void_p(d_depth_ptr), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"rsi_synth_generate_device failed: {rc}")
    del keep


def write_case_files(lib, fasta, depth, workdir, chrom="chrS"):
    """FASTA (+ .fai) and "pos depth" text of one chromosome for command-line runs; returns (fasta path, depth path)."""
    import os
    fa, rd = os.path.join(workdir, "ref.fa"), os.path.join(workdir, "depth.txt")
    lib.rsi_synth_write_fasta.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.rsi_synth_write_depth_text.argtypes = [C.c_char_p, C.c_void_p, C.c_int64]
    fasta = np.ascontiguousarray(fasta, dtype=np.uint8)
    depth = np.ascontiguousarray(depth, dtype=np.int32)
    if lib.rsi_synth_write_fasta(fa.encode(), chrom.encode(), fasta.ctypes.data, fasta.size) != 0:
        raise RuntimeError("rsi_synth_write_fasta failed")
    if lib.rsi_synth_write_depth_text(rd.encode(), depth.ctypes.data, depth.size) != 0:
        raise RuntimeError("rsi_synth_write_depth_text failed")
    return fa, rd
