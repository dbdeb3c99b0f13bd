"""ctypes binding of librsi_hot.so (include/rsi_hot.h) -- the drop-in for the reference's
per-chromosome hot path (rsi.cpp:2189-2212).

The library is the product: there is no Python or CPU fallback.  `load_library()` raises if the
shared object is missing, and `RsiHot()` raises if no HIP device can be opened.
"""
import ctypes as C
import os

import numpy as np

# A pool drives one HIP stream per worker.  The ROCm runtime multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4); with a dozen streams that puts unrelated chromosomes
# in line behind each other's kernels (measured: ~10 % of the genome rate).  The variable is read when
# the HIP runtime initialises, so it has to be in the environment before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSI_HOT_LIB: another build of the same ABI (A/B measurements of library versions, tools/ab_bench.py)
LIB_PATH = os.environ.get("RSI_HOT_LIB") or os.path.join(_HERE, "librsi_hot.so")

RSI_OK = 0
STATUS_NAMES = {0: "RSI_OK", -1: "RSI_ERR_NO_DEVICE", -2: "RSI_ERR_BAD_ARG", -3: "RSI_ERR_HIP",
                -4: "RSI_ERR_TOO_SMALL", -5: "RSI_ERR_UNSUPPORTED", -6: "RSI_ERR_INTERNAL"}

# every symbol include/rsi_hot.h and include/rsi_synth.h declare
EXPORTS = ["rsi_default_params", "rsi_hot_create", "rsi_hot_destroy", "rsi_hot_last_error", "rsi_hot_run",
           "rsi_hot_run_device", "rsi_hot_load_depth_text", "rsi_hot_run_text", "rsi_hot_load_depth_bam", "rsi_hot_run_bam", "rsi_bam_references", "rsi_result_annotate_bam", "rsi_result_summary", "rsi_summary_format_row", "rsi_summary_format_rows", "rsi_result_pairs", "rsi_result_ncalls", "rsi_result_calls", "rsi_result_stats", "rsi_result_noncode",
           "rsi_result_format_row", "rsi_result_free", "rsi_hot_fetch_i32", "rsi_hot_fetch_f32", "rsi_hot_fetch_i64",
           "rsi_hot_kernel_times", "rsi_hot_phase_times", "rsi_hot_set_timing", "rsi_pool_create", "rsi_pool_destroy", "rsi_pool_workers", "rsi_pool_worker",
           "rsi_pool_set_timing", "rsi_pool_set_timing_kernel", "rsi_hot_set_timing_kernel", "rsi_pool_set_schedule", "rsi_pool_last_error", "rsi_pool_run", "rsi_pool_run_host", "rsi_pool_submit", "rsi_pool_wait", "rsi_plot_expand", "rsi_plot_write_files", "rsi_result_log_line", "rsi_hot_debug_level_sums", "rsi_hot_debug_scan", "rsi_synth_generate_host", "rsi_synth_generate_device", "rsi_synth_write_depth_text", "rsi_synth_write_fasta"]


class RsiParams(C.Structure):
    _fields_ = [("m", C.c_int32), ("gcadjust", C.c_int32), ("trans", C.c_int32), ("merge", C.c_int32),
                ("maxchkbp", C.c_int32), ("debug", C.c_int32), ("cap", C.c_double), ("epsilon", C.c_double),
                ("threshold", C.c_double), ("chklen", C.c_double), ("minmlen", C.c_double),
                ("buffer", C.c_double), ("p", C.c_double)]


class RsiCall(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("type", C.c_int32), ("geno", C.c_int32),
                ("status", C.c_int32), ("length", C.c_int32), ("qscore", C.c_int32), ("pad", C.c_int32),
                ("score", C.c_double), ("p1", C.c_double), ("cnvmed", C.c_double), ("cnvsd", C.c_double),
                ("cnviqr", C.c_double), ("refmed", C.c_double), ("refsd", C.c_double), ("refiqr", C.c_double)]


class RsiChromStats(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_compact", C.c_int64), ("nbins", C.c_int64), ("n_noncode", C.c_int32),
                ("Lmax", C.c_int32), ("gc_rdmean", C.c_double), ("cap_median", C.c_double), ("RDmedian", C.c_double),
                ("RDsd", C.c_double), ("nb_mad", C.c_double), ("nb_r", C.c_double), ("nb_tmin", C.c_double),
                ("tmedian1", C.c_double), ("tsigma1", C.c_double), ("tlamda1", C.c_double), ("tmedian2", C.c_double),
                ("tsigma2", C.c_double), ("tlamda2", C.c_double), ("trim_escapes", C.c_int32),
                ("inexact_sums", C.c_int32), ("t_device_ms", C.c_double), ("t_kernels_ms", C.c_double),
                ("byte_escapes", C.c_int64), ("scan_tiles", C.c_int32), ("scan_tiles_listed", C.c_int32)]


RSI_MAX_TIMED = 64
SUMMARY_HEAD, SUMMARY_CALL = 8, 8   # rsi_hot.h: RSI_SUMMARY_HEAD, RSI_SUMMARY_CALL


class RsiTextStats(C.Structure):
    _fields_ = [("bytes", C.c_int64), ("lines", C.c_int64), ("stored", C.c_int64), ("beyond", C.c_int64),
                ("fallback", C.c_int32), ("pad", C.c_int32), ("t_total_ms", C.c_double), ("t_parse_kernel_ms", C.c_double)]


class RsiBamStats(C.Structure):
    _fields_ = [("n", C.c_int64), ("bytes_compressed", C.c_int64), ("bytes_inflated", C.c_int64), ("records", C.c_int64), ("on_chrom", C.c_int64),
                ("used", C.c_int64), ("runs", C.c_int64), ("tid", C.c_int32), ("indexed", C.c_int32),
                ("t_total_ms", C.c_double), ("t_inflate_ms", C.c_double), ("t_walk_ms", C.c_double), ("t_wait_ms", C.c_double),
                ("malformed", C.c_int64)]


class RsiBatchTimes(C.Structure):
    _fields_ = [("nkernels", C.c_int32), ("nphases", C.c_int32), ("kernel_name", C.c_char_p * RSI_MAX_TIMED),
                ("kernel_ms", C.c_double * RSI_MAX_TIMED), ("kernel_launches", C.c_int64 * RSI_MAX_TIMED),
                ("kernel_bases", C.c_int64 * RSI_MAX_TIMED), ("phase_name", C.c_char_p * RSI_MAX_TIMED),
                ("phase_ms", C.c_double * RSI_MAX_TIMED)]


CALL_FIELDS = [f[0] for f in RsiCall._fields_ if f[0] != "pad"]
WHICH = {"calls": 0, "calls_raw": 1, "segs": 2, "blocks": 3}

_lib = None


def load_library():
    """dlopen librsi_hot.so; raises (never falls back) when the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -f rsicnv_amd/csrc/Makefile` "
                           "(or __graft_entry__.build()); rsicnv_amd has no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64;
    # if librsi_hot.so were loaded first it would pull /opt/rocm's copy and the second runtime to
    # touch the GPU would fail with "no ROCm-capable device".  Importing torch first makes our
    # DT_NEEDED libamdhip64.so.7 resolve to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.rsi_default_params.argtypes = [C.POINTER(RsiParams)]
    L.rsi_hot_create.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.rsi_hot_create.restype = C.c_void_p
    L.rsi_hot_destroy.argtypes = [C.c_void_p]
    L.rsi_hot_last_error.argtypes = [C.c_void_p]
    L.rsi_hot_last_error.restype = C.c_char_p
    L.rsi_hot_run.argtypes = [C.c_void_p, C.POINTER(RsiParams), C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    L.rsi_hot_run_device.argtypes = [C.c_void_p, C.POINTER(RsiParams), C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    L.rsi_hot_load_depth_text.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(RsiTextStats)]
    L.rsi_hot_run_text.argtypes = [C.c_void_p, C.POINTER(RsiParams), C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(RsiTextStats)]
    L.rsi_hot_load_depth_bam.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(RsiBamStats)]
    L.rsi_hot_run_bam.argtypes = [C.c_void_p, C.POINTER(RsiParams), C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                  C.POINTER(C.c_void_p), C.POINTER(RsiBamStats)]
    L.rsi_result_ncalls.argtypes = [C.c_void_p, C.c_int]
    L.rsi_result_calls.argtypes = [C.c_void_p, C.c_int]
    L.rsi_result_calls.restype = C.POINTER(RsiCall)
    L.rsi_result_stats.argtypes = [C.c_void_p]
    L.rsi_result_stats.restype = C.POINTER(RsiChromStats)
    L.rsi_result_noncode.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int]
    L.rsi_result_summary.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.rsi_result_summary.restype = C.c_int
    L.rsi_summary_format_row.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    L.rsi_summary_format_row.restype = C.c_int
    L.rsi_summary_format_rows.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]
    L.rsi_summary_format_rows.restype = C.c_int
    L.rsi_result_log_line.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    L.rsi_result_log_line.restype = C.c_int
    L.rsi_result_format_row.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int]
    L.rsi_result_free.argtypes = [C.c_void_p]
    for nm, ct in (("rsi_hot_fetch_i32", C.c_int32), ("rsi_hot_fetch_f32", C.c_float), ("rsi_hot_fetch_i64", C.c_int64)):
        f = getattr(L, nm)
        f.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(ct), C.c_int64]
        f.restype = C.c_int64
    L.rsi_hot_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]
    L.rsi_hot_phase_times.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.c_int]
    L.rsi_hot_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.rsi_pool_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.rsi_pool_create.restype = C.c_void_p
    L.rsi_pool_destroy.argtypes = [C.c_void_p]
    L.rsi_pool_workers.argtypes = [C.c_void_p]
    L.rsi_pool_worker.argtypes = [C.c_void_p, C.c_int]
    L.rsi_pool_worker.restype = C.c_void_p
    L.rsi_pool_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.rsi_pool_set_timing_kernel.argtypes = [C.c_void_p, C.c_char_p]
    L.rsi_hot_set_timing_kernel.argtypes = [C.c_void_p, C.c_char_p]
    L.rsi_pool_set_schedule.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.rsi_pool_last_error.argtypes = [C.c_void_p]
    L.rsi_pool_last_error.restype = C.c_char_p
    L.rsi_pool_run.argtypes = [C.c_void_p, C.POINTER(RsiParams), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                               C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(RsiBatchTimes)]
    L.rsi_pool_run_host.argtypes = [C.c_void_p, C.POINTER(RsiParams), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                               C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(RsiBatchTimes)]
    L.rsi_pool_submit.argtypes = L.rsi_pool_run.argtypes
    L.rsi_pool_submit.restype = C.c_uint64
    L.rsi_pool_wait.argtypes = [C.c_void_p, C.c_uint64]
    _lib = L
    return L


def make_params(m=101, gcadjust=1, trans=0, merge=1, maxchkbp=100000, debug=0, cap=4.0, epsilon=1.5,
                threshold=-1.0, chklen=2.5, minmlen=3.01, buffer=0.05, p=0.05):
    """rsi:: defaults (rsi.cpp:34-98); m is forced odd as get_parameters does (rsi.cpp:2061-2064).
    trans: 0 NBN (-NB), 1 MED (-MED), 2 ALL (-ALL)."""
    if m % 2 != 1:
        m += 1
    return RsiParams(m, gcadjust, trans, merge, maxchkbp, debug, cap, epsilon, threshold, chklen, minmlen, buffer, p)


class RsiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


class Result:
    """One chromosome's results.  The C object is kept and read on demand (list by list), so that a caller
    that only wants the final calls does not pay for converting thousands of intermediate segments."""

    def __init__(self, lib, handle):
        self._lib = lib
        self._h = handle
        self._lists = {}
        self._rows = None
        self._noncode = None
        self._stats = None

    def __del__(self):
        try:
            if self._h:
                self._lib.rsi_result_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def stats(self):
        if self._stats is None:
            st = self._lib.rsi_result_stats(self._h).contents
            self._stats = {f[0]: getattr(st, f[0]) for f in RsiChromStats._fields_}
        return self._stats

    def calls(self, which="calls"):
        if which not in self._lists:
            w = WHICH[which]
            k = self._lib.rsi_result_ncalls(self._h, w)
            arr = self._lib.rsi_result_calls(self._h, w)
            self._lists[which] = [{f: getattr(arr[i], f) for f in CALL_FIELDS} for i in range(k)]
        return self._lists[which]

    def summary_into(self, row, chrom_id, max_calls):
        """Write the fixed-layout summary block (rsi_result_summary: [chrom id, median, SD, number of calls, stored, 0, 0, 0]
        then 8 doubles per stored call) into a float64 numpy row; returns the number of doubles written."""
        assert row.dtype == np.float64 and row.flags["C_CONTIGUOUS"] and row.size >= SUMMARY_HEAD + SUMMARY_CALL * max_calls
        return self._lib.rsi_result_summary(self._h, int(chrom_id), row.ctypes.data, int(max_calls))

    def log_lines(self):
        """The per-L "DEL-" / "DUP+" lines of the scan passes (rsi.cpp:1221-1224, 1251-1254)."""
        out, i, buf = [], 0, C.create_string_buffer(256)
        while True:
            i = self._lib.rsi_result_log_line(self._h, i, buf, 256)
            if i <= 0:
                return out
            out.append(buf.value.decode())

    @property
    def lists(self):
        return {name: self.calls(name) for name in WHICH}

    @property
    def noncode(self):
        if self._noncode is None:
            k = self._lib.rsi_result_noncode(self._h, None, 0)
            pairs = (C.c_int32 * max(2 * k, 2))()
            self._lib.rsi_result_noncode(self._h, pairs, k)
            self._noncode = np.array(pairs[:2 * k], dtype=np.int32)
        return self._noncode

    @property
    def rows(self):
        if self._rows is None:
            buf = C.create_string_buffer(1024)
            self._rows = []
            for i in range(self._lib.rsi_result_ncalls(self._h, WHICH["calls"])):
                self._lib.rsi_result_format_row(self._h, i, b"%CHROM%", buf, 1024)
                self._rows.append(buf.value.decode())
        return self._rows

    def format_rows(self, chrom):
        return [r.replace("%CHROM%", chrom) for r in self.rows]


class RsiHot:
    """One context = one GPU + one stream (rsi_hot_create)."""

    def __init__(self, device=0):
        self.lib = load_library()
        st = C.c_int(0)
        self.ctx = self.lib.rsi_hot_create(device, C.byref(st))
        if not self.ctx:
            raise RsiError(st.value, self.lib.rsi_hot_last_error(None).decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.rsi_hot_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != RSI_OK:
            raise RsiError(rc, self.lib.rsi_hot_last_error(self.ctx).decode())

    def set_timing(self, on=True):
        self.lib.rsi_hot_set_timing(self.ctx, int(on))

    def debug_level_sums(self, T, status, Lmax):
        """filterstatus' per-level float sums of host arrays on the device (test hook): (sums, counts), index = level + Lmax."""
        t = np.ascontiguousarray(T, dtype=np.float32)
        s = np.ascontiguousarray(status, dtype=np.int32)
        assert t.size == s.size
        sums = np.zeros(2 * Lmax + 1, dtype=np.float32)
        counts = np.zeros(2 * Lmax + 1, dtype=np.int32)
        self.lib.rsi_hot_debug_level_sums.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
        self._check(self.lib.rsi_hot_debug_level_sums(self.ctx, t.ctypes.data, s.ctypes.data, t.size, int(Lmax), sums.ctypes.data, counts.ctypes.data))
        return sums, counts

    def debug_scan(self, T, medint, RDmedian, tmedian, tlamda, Lmax):
        """One scan pass (rsistatus) over host arrays on the device (test hook): (status, info)."""
        t = np.ascontiguousarray(T, dtype=np.float32)
        mi = np.ascontiguousarray(medint, dtype=np.int32)
        assert t.size == mi.size
        st = np.zeros(t.size, dtype=np.int32)
        info = np.zeros(4, dtype=np.int32)
        self.lib.rsi_hot_debug_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int,
                                                C.c_void_p, C.c_void_p]
        self._check(self.lib.rsi_hot_debug_scan(self.ctx, t.ctypes.data, mi.ctypes.data, t.size, float(RDmedian), float(tmedian), float(tlamda),
                                                int(Lmax), st.ctypes.data, info.ctypes.data))
        return st, info

    def run(self, params, depth, fasta):
        """depth: int32[n] raw per-base depth, fasta: uint8[n] sequence bytes (host arrays)."""
        d = np.ascontiguousarray(depth, dtype=np.int32)
        f = np.ascontiguousarray(fasta, dtype=np.uint8)
        if d.shape != f.shape or d.ndim != 1:
            raise ValueError("depth and fasta must be 1-D arrays of the same length")
        out = C.c_void_p()
        self._check(self.lib.rsi_hot_run(self.ctx, C.byref(params), d.ctypes.data, f.ctypes.data, d.size, C.byref(out)))
        return Result(self.lib, out)

    def load_depth_text(self, path, n):
        """Parse a "pos depth" text file into the context's device depth buffer (rsi_hot_load_depth_text).
        Returns the statistics as a dict; fetch("depth_in") reads the result back."""
        st = RsiTextStats()
        self._check(self.lib.rsi_hot_load_depth_text(self.ctx, os.fsencode(path), int(n), C.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in RsiTextStats._fields_ if f[0] != "pad"}

    def run_text(self, params, path, fasta):
        """Depth from a text file (parsed on the device), fasta: uint8[n] host array."""
        f = np.ascontiguousarray(fasta, dtype=np.uint8)
        out = C.c_void_p()
        st = RsiTextStats()
        self._check(self.lib.rsi_hot_run_text(self.ctx, C.byref(params), os.fsencode(path), f.ctypes.data, f.size, C.byref(out), C.byref(st)))
        res = Result(self.lib, out)
        res.text_stats = {f_[0]: getattr(st, f_[0]) for f_ in RsiTextStats._fields_ if f_[0] != "pad"}
        return res

    def load_depth_bam(self, bam, chrom, minq=0, min_baseq=13):
        """Per-base depth of `chrom` from a BAM file into the context's device depth buffer (rsi_hot_load_depth_bam)."""
        st = RsiBamStats()
        self._check(self.lib.rsi_hot_load_depth_bam(self.ctx, os.fsencode(bam), chrom.encode(), int(minq), int(min_baseq), C.byref(st)))
        return {f[0]: getattr(st, f[0]) for f in RsiBamStats._fields_}

    def run_bam(self, params, bam, chrom, fasta, minq=0, min_baseq=13):
        f = np.ascontiguousarray(fasta, dtype=np.uint8)
        out = C.c_void_p()
        st = RsiBamStats()
        self._check(self.lib.rsi_hot_run_bam(self.ctx, C.byref(params), os.fsencode(bam), chrom.encode(), int(minq), int(min_baseq),
                                             f.ctypes.data, f.size, C.byref(out), C.byref(st)))
        res = Result(self.lib, out)
        res.bam_stats = {f_[0]: getattr(st, f_[0]) for f_ in RsiBamStats._fields_}
        return res

    def run_device(self, params, d_depth_ptr, d_fasta_ptr, n):
        """Inputs already in HBM (raw device pointers, 16-byte aligned)."""
        out = C.c_void_p()
        self._check(self.lib.rsi_hot_run_device(self.ctx, C.byref(params), C.c_void_p(d_depth_ptr), C.c_void_p(d_fasta_ptr),
                                                n, C.byref(out)))
        return Result(self.lib, out)

    def fetch(self, name):
        for fn, ct, dt in ((self.lib.rsi_hot_fetch_i32, C.c_int32, np.int32), (self.lib.rsi_hot_fetch_f32, C.c_float, np.float32),
                           (self.lib.rsi_hot_fetch_i64, C.c_int64, np.int64)):
            n = fn(self.ctx, name.encode(), None, 0)
            if n >= 0:
                out = np.zeros(n, dtype=dt)
                k = fn(self.ctx, name.encode(), out.ctypes.data_as(C.POINTER(ct)), n)
                if k < 0:
                    raise RsiError(int(k), self.lib.rsi_hot_last_error(self.ctx).decode())
                return out
        raise KeyError(name)

    def phase_times(self):
        names = (C.c_char_p * 64)()
        ms = (C.c_double * 64)()
        k = self.lib.rsi_hot_phase_times(self.ctx, names, ms, 64)
        return [(names[i].decode(), float(ms[i])) for i in range(min(k, 64))]

    def kernel_times(self):
        names = (C.c_char_p * 4096)()
        ms = (C.c_float * 4096)()
        k = self.lib.rsi_hot_kernel_times(self.ctx, names, ms, 4096)
        return [(names[i].decode(), float(ms[i])) for i in range(min(k, 4096))]


class RsiPool:
    """Several chromosomes in flight on one GPU (rsi_pool_*): `workers` host threads, each with its
    own stream and workspace; the HBM-bound per-base phase is taken in turns."""

    def __init__(self, device=0, workers=8):
        self.lib = load_library()
        st = C.c_int(0)
        self.pool = self.lib.rsi_pool_create(device, workers, C.byref(st))
        if not self.pool:
            raise RsiError(st.value, self.lib.rsi_hot_last_error(None).decode())
        self.times = RsiBatchTimes()

    def close(self):
        if getattr(self, "pool", None):
            self.lib.rsi_pool_destroy(self.pool)
            self.pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_timing(self, on=True):
        self.lib.rsi_pool_set_timing(self.pool, int(on))

    def set_timing_kernel(self, name):
        """The kernel timing mode 3 brackets (a name of kernel_table())."""
        self.lib.rsi_pool_set_timing_kernel(self.pool, name.encode())

    def set_schedule(self, isolate=False, streamers=0):
        """isolate: per-base phases run alone on the chip (clean kernel timings); streamers: per-base phases in flight (0 = keep)."""
        self.lib.rsi_pool_set_schedule(self.pool, 1 if isolate else 0, int(streamers))

    def reset_times(self):
        self.times = RsiBatchTimes()

    def run(self, params, chroms, collect_times=False, host=False):
        """chroms: list of (d_depth_ptr, d_fasta_ptr, n): device pointers, or with host=True host pointers (pinned for
        asynchronous transfers).  Returns a list of Result in input order."""
        k = len(chroms)
        dp = (C.c_void_p * k)(*[C.c_void_p(c[0]) for c in chroms])
        fp = (C.c_void_p * k)(*[C.c_void_p(c[1]) for c in chroms])
        nn = (C.c_int64 * k)(*[c[2] for c in chroms])
        out = (C.c_void_p * k)()
        st = (C.c_int * k)()
        fn = self.lib.rsi_pool_run_host if host else self.lib.rsi_pool_run
        rc = fn(self.pool, C.byref(params), k, dp, fp, nn, out, st, C.byref(self.times) if collect_times else None)
        if rc != RSI_OK:
            for i in range(k):
                if out[i]:
                    self.lib.rsi_result_free(out[i])
            raise RsiError(rc, self.lib.rsi_pool_last_error(self.pool).decode())
        return [Result(self.lib, C.c_void_p(out[i])) for i in range(k)]

    def submit(self, params, chroms, collect_times=False):
        """Queue a run (rsi_pool_submit) and return a handle for wait(): the chromosomes of queued runs go to the workers in
        submission order, so consecutive samples overlap.  chroms as for run() (device pointers)."""
        k = len(chroms)
        h = {"k": k, "params": params,
             "dp": (C.c_void_p * k)(*[C.c_void_p(c[0]) for c in chroms]), "fp": (C.c_void_p * k)(*[C.c_void_p(c[1]) for c in chroms]),
             "nn": (C.c_int64 * k)(*[c[2] for c in chroms]), "out": (C.c_void_p * k)(), "st": (C.c_int * k)()}
        h["times"] = self.times if collect_times else None   # the run writes into it when it finishes: alive as long as the handle
        h["ticket"] = self.lib.rsi_pool_submit(self.pool, C.byref(params), k, h["dp"], h["fp"], h["nn"], h["out"], h["st"],
                                               C.byref(h["times"]) if collect_times else None)
        if not h["ticket"]:
            raise RsiError(-2, "rsi_pool_submit: bad arguments")
        return h

    def wait(self, h):
        """The results of a submitted run, in input order (the calling thread works as one of the pool's workers meanwhile)."""
        rc = self.lib.rsi_pool_wait(self.pool, h["ticket"])
        k, out = h["k"], h["out"]
        if rc != RSI_OK:
            for i in range(k):
                if out[i]:
                    self.lib.rsi_result_free(out[i])
            raise RsiError(rc, self.lib.rsi_pool_last_error(self.pool).decode())
        return [Result(self.lib, C.c_void_p(out[i])) for i in range(k)]

    def kernel_table(self):
        t = self.times
        return {t.kernel_name[i].decode(): (t.kernel_ms[i], t.kernel_launches[i], t.kernel_bases[i]) for i in range(t.nkernels)}

    def phase_table(self):
        t = self.times
        return {t.phase_name[i].decode(): t.phase_ms[i] for i in range(t.nphases)}
