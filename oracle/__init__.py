"""ctypes bindings for the CPU checkers (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (rsicnv_amd/) never does.

* ``Oracle``  -- oracle/librsi_oracle.so, the CPU restatement (oracle/rsi_oracle.cpp).
* ``Ref``     -- oracle/_ref/libref.so, the real reference compiled by oracle/Makefile.ref with
                 the stage driver oracle/ref_driver.cpp.  Single global session (the reference
                 keeps its state in globals), so use one instance at a time.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "librsi_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref.so")
REF_BIN = os.path.join(_HERE, "_ref", "rsicnv_ref")


class Params(C.Structure):
    """Same layout as orc_params / ref_params."""
    _fields_ = [("m", C.c_int32), ("gcadjust", C.c_int32), ("trans", C.c_int32), ("merge", C.c_int32),
                ("maxchkbp", C.c_int32), ("debug", C.c_int32), ("cap", C.c_double), ("epsilon", C.c_double),
                ("threshold", C.c_double), ("chklen", C.c_double), ("minmlen", C.c_double),
                ("buffer", C.c_double), ("p", C.c_double)]


class Call(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("type", C.c_int32), ("geno", C.c_int32),
                ("status", C.c_int32), ("length", C.c_int32), ("qscore", C.c_int32), ("pad", C.c_int32),
                ("score", C.c_double), ("p1", C.c_double), ("cnvmed", C.c_double), ("cnvsd", C.c_double),
                ("cnviqr", C.c_double), ("refmed", C.c_double), ("refsd", C.c_double), ("refiqr", C.c_double)]


class ScanScalars(C.Structure):
    _fields_ = [("tmedian1", C.c_double), ("tsigma1", C.c_double), ("tlamda1", C.c_double),
                ("tmedian2", C.c_double), ("tsigma2", C.c_double), ("tlamda2", C.c_double),
                ("target_tlamda", C.c_double), ("Lmax", C.c_int32), ("cal_max", C.c_int32),
                ("stepwise_matches_reference", C.c_int32), ("nseg", C.c_int32)]


CALL_FIELDS = [f[0] for f in Call._fields_ if f[0] != "pad"]


def calls_to_dicts(arr, n):
    return [{k: getattr(arr[i], k) for k in CALL_FIELDS} for i in range(n)]


def make_params(m=101, gcadjust=1, trans=0, merge=1, maxchkbp=100000, debug=0, cap=4.0, epsilon=1.5,
                threshold=-1.0, chklen=2.5, minmlen=3.01, buffer=0.05, p=0.05):
    """Reference defaults (rsi.cpp:34-98); trans: 0 NBN, 1 MED, 2 ALL.  m is forced odd as
    get_parameters does (rsi.cpp:2061-2064)."""
    if m % 2 != 1:
        m += 1
    return Params(m, gcadjust, trans, merge, maxchkbp, debug, cap, epsilon, threshold, chklen, minmlen, buffer, p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run `make -f oracle/Makefile` (or __graft_entry__.build())")
        L = self.lib = C.CDLL(ORACLE_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_run.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_int32, C.c_int32]
        L.orc_run.restype = C.c_int
        for nm, ct in (("orc_get_i32", C.c_int32), ("orc_get_f32", C.c_float), ("orc_get_f64", C.c_double)):
            f = getattr(L, nm)
            f.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(ct), C.c_int64]
            f.restype = C.c_int64
        L.orc_get_calls.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Call), C.c_int32]
        L.orc_get_calls.restype = C.c_int
        for nm, ct in (("orc_median_i32", C.c_int32), ("orc_median_f32", C.c_float), ("orc_median_f64", C.c_double),
                       ("orc_iqr_i32", C.c_int32), ("orc_iqr_f32", C.c_float), ("orc_exact_median_i32", C.c_int32)):
            f = getattr(L, nm)
            f.argtypes = [C.POINTER(ct), C.c_int64]
            f.restype = C.c_double
        L.orc_pnorm.argtypes = [C.c_double]
        L.orc_pnorm.restype = C.c_double
        self.h = C.c_void_p(L.orc_create())

    def __del__(self):
        try:
            self.lib.orc_destroy(self.h)
        except Exception:
            pass

    def run(self, params, depth, fasta, snapshots=True):
        d, dp = _i32(depth)
        f, fp = _u8(fasta)
        assert d.shape == f.shape
        return self.lib.orc_run(self.h, C.byref(params), dp, fp, d.size, 1 if snapshots else 0)

    def _get(self, fn, ct, dt, name):
        n = fn(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.zeros(n, dtype=dt)
        fn(self.h, name.encode(), out.ctypes.data_as(C.POINTER(ct)), n)
        return out

    def i32(self, name):
        return self._get(self.lib.orc_get_i32, C.c_int32, np.int32, name)

    def f32(self, name):
        return self._get(self.lib.orc_get_f32, C.c_float, np.float32, name)

    def f64(self, name):
        return self._get(self.lib.orc_get_f64, C.c_double, np.float64, name)

    def calls(self, which="calls"):
        n = self.lib.orc_get_calls(self.h, which.encode(), None, 0)
        arr = (Call * max(n, 1))()
        self.lib.orc_get_calls(self.h, which.encode(), arr, n)
        return calls_to_dicts(arr, n)

    # numeric-utility probes
    def median(self, x):
        x = np.ascontiguousarray(x)
        fn, ct = {np.dtype(np.int32): (self.lib.orc_median_i32, C.c_int32),
                  np.dtype(np.float32): (self.lib.orc_median_f32, C.c_float),
                  np.dtype(np.float64): (self.lib.orc_median_f64, C.c_double)}[x.dtype]
        return fn(x.ctypes.data_as(C.POINTER(ct)), x.size)

    def iqr(self, x):
        x = np.ascontiguousarray(x)
        fn, ct = {np.dtype(np.int32): (self.lib.orc_iqr_i32, C.c_int32),
                  np.dtype(np.float32): (self.lib.orc_iqr_f32, C.c_float)}[x.dtype]
        return fn(x.ctypes.data_as(C.POINTER(ct)), x.size)

    def exact_median(self, x):
        x = np.ascontiguousarray(x, dtype=np.int32)
        return self.lib.orc_exact_median_i32(x.ctypes.data_as(C.POINTER(C.c_int32)), x.size)

    def pnorm(self, x):
        return self.lib.orc_pnorm(x)


def ref_available():
    return os.path.exists(REF_SO)


class Ref:
    """The compiled reference behind oracle/ref_driver.cpp."""

    def __init__(self):
        if not ref_available():
            raise RuntimeError(f"{REF_SO} missing: needs /root/reference and `make -f oracle/Makefile.ref`")
        L = self.lib = C.CDLL(REF_SO)
        L.ref_load.argtypes = [C.POINTER(Params), C.POINTER(C.c_int32), C.c_char_p, C.c_int32, C.c_char_p, C.c_char_p]
        L.ref_get_rd.argtypes = [C.POINTER(C.c_int32), C.c_int32]
        L.ref_get_noncode.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32]
        L.ref_get_chrom_scalars.argtypes = [C.POINTER(C.c_double)]
        L.ref_get_bins.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int32]
        L.ref_scan_stepwise.argtypes = [C.c_int, C.POINTER(ScanScalars)]
        L.ref_get_status.argtypes = [C.c_int, C.POINTER(C.c_int32), C.c_int32]
        L.ref_get_segs.argtypes = [C.POINTER(Call), C.c_int32]
        L.ref_stage_blocks.argtypes = [C.POINTER(Call), C.c_int32]
        L.ref_get_calls.argtypes = [C.c_int, C.POINTER(Call), C.c_int32]
        L.ref_format_calls.argtypes = [C.c_char_p, C.c_int32]
        L.ref_run_timed.argtypes = [C.POINTER(Params), C.POINTER(C.c_int32), C.c_char_p, C.c_int32, C.POINTER(C.c_double)]
        for nm, ct in (("ref_median_i32", C.c_int32), ("ref_median_f32", C.c_float), ("ref_median_f64", C.c_double),
                       ("ref_iqr_i32", C.c_int32), ("ref_iqr_f32", C.c_float), ("ref_exact_median_i32", C.c_int32)):
            f = getattr(L, nm)
            f.argtypes = [C.POINTER(ct), C.c_int64]
            f.restype = C.c_double
        L.ref_pnorm.argtypes = [C.c_double]
        L.ref_pnorm.restype = C.c_double
        L.ref_runmean_f32.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_int32]

    def load(self, params, depth, fasta, chrom="chrS", log=None):
        d, dp = _i32(depth)
        f = np.ascontiguousarray(fasta, dtype=np.uint8)
        self._keep = (d, f)
        return self.lib.ref_load(C.byref(params), dp, f.ctypes.data_as(C.c_char_p), d.size, chrom.encode(),
                                 (log or "").encode())

    def stage_gc(self):
        return self.lib.ref_stage_gc()

    def stage_cap(self):
        return self.lib.ref_stage_cap()

    def stage_concat(self):
        return self.lib.ref_stage_concat()

    def rd(self):
        n = self.lib.ref_rd_size()
        out = np.zeros(n, dtype=np.int32)
        self.lib.ref_get_rd(out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def noncode(self):
        n = self.lib.ref_get_noncode(None, None, 0)
        b = np.zeros(max(n, 1), dtype=np.int32)
        e = np.zeros(max(n, 1), dtype=np.int32)
        self.lib.ref_get_noncode(b.ctypes.data_as(C.POINTER(C.c_int32)), e.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return np.stack([b[:n], e[:n]], axis=1).reshape(-1)

    def chrom_scalars(self):
        out = (C.c_double * 2)()
        self.lib.ref_get_chrom_scalars(out)
        return out[0], out[1]

    def stage_bins(self):
        nb = self.lib.ref_stage_bins()
        med = np.zeros(nb, dtype=np.float32)
        medint = np.zeros(nb, dtype=np.int32)
        nbn = np.zeros(nb, dtype=np.float32)
        self.lib.ref_get_bins(med.ctypes.data_as(C.POINTER(C.c_float)), medint.ctypes.data_as(C.POINTER(C.c_int32)),
                              nbn.ctypes.data_as(C.POINTER(C.c_float)), nb)
        return med, medint, nbn

    def scan(self, use_med=False):
        sc = ScanScalars()
        nb = self.lib.ref_scan_stepwise(1 if use_med else 0, C.byref(sc))
        st = []
        for w in range(3):
            a = np.zeros(nb, dtype=np.int32)
            self.lib.ref_get_status(w, a.ctypes.data_as(C.POINTER(C.c_int32)), nb)
            st.append(a)
        n = self.lib.ref_get_segs(None, 0)
        arr = (Call * max(n, 1))()
        self.lib.ref_get_segs(arr, n)
        return sc, st, calls_to_dicts(arr, n)

    def blocks(self):
        arr = (Call * 65536)()
        n = self.lib.ref_stage_blocks(arr, 65536)
        return calls_to_dicts(arr, n)

    def detect(self):
        self.lib.ref_stage_detect()
        out = []
        for filt in (0, 1):
            n = self.lib.ref_get_calls(filt, None, 0)
            arr = (Call * max(n, 1))()
            self.lib.ref_get_calls(filt, arr, n)
            out.append(calls_to_dicts(arr, n))
        buf = C.create_string_buffer(1 << 20)
        k = self.lib.ref_format_calls(buf, 1 << 20)
        return out[0], out[1], buf.value.decode() if k >= 0 else None

    def run_timed(self, params, depth, fasta):
        d, dp = _i32(depth)
        f = np.ascontiguousarray(fasta, dtype=np.uint8)
        st = (C.c_double * 5)()
        nc = self.lib.ref_run_timed(C.byref(params), dp, f.ctypes.data_as(C.c_char_p), d.size, st)
        return nc, list(st)

    def median(self, x):
        x = np.ascontiguousarray(x)
        fn, ct = {np.dtype(np.int32): (self.lib.ref_median_i32, C.c_int32),
                  np.dtype(np.float32): (self.lib.ref_median_f32, C.c_float),
                  np.dtype(np.float64): (self.lib.ref_median_f64, C.c_double)}[x.dtype]
        return fn(x.ctypes.data_as(C.POINTER(ct)), x.size)

    def iqr(self, x):
        x = np.ascontiguousarray(x)
        fn, ct = {np.dtype(np.int32): (self.lib.ref_iqr_i32, C.c_int32),
                  np.dtype(np.float32): (self.lib.ref_iqr_f32, C.c_float)}[x.dtype]
        return fn(x.ctypes.data_as(C.POINTER(ct)), x.size)

    def exact_median(self, x):
        x = np.ascontiguousarray(x, dtype=np.int32)
        return self.lib.ref_exact_median_i32(x.ctypes.data_as(C.POINTER(C.c_int32)), x.size)

    def pnorm(self, x):
        return self.lib.ref_pnorm(x)

    def runmean(self, y, band):
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = np.zeros_like(y)
        self.lib.ref_runmean_f32(y.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)), y.size, band)
        return out
