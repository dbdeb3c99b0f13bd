/* rsi_hot.h -- C ABI of the MI355X-native RSI read-depth CNV hot path (librsi_hot.so).
 *
 * The reference (yhwu/rsicnv) has no plugin / FFI interface; its hot path sits behind a seam of
 * free functions operating on `Array<int>& RD` plus `rsi::` globals, called from main's
 * per-chromosome loop (rsi.cpp:2189-2217).  One rsi_hot_run() call replaces, for one chromosome:
 *
 *   load_data_from_text, after its parse loop        loaddata.cpp:478-486, 519-531
 *     GC mask + get_noseq_regions                    loaddata.cpp:481-486, 243-273; readref.cpp:88
 *     checkgccontent / adjustgccontent               gccontent.cpp:95, 43
 *     apply_cap                                      loaddata.cpp:229
 *   concatenate_data                                 loaddata.cpp:48      (rsi.cpp:2200)
 *   rsi::RDmedian = _median(RD); rsi::RDsd = ...     rsi.cpp:2202-2203    (wufunctions.cpp:364, 766)
 *   detectcnv                                        rsi.cpp:1795         (rsi.cpp:2206)
 *   sd_filters                                       rsi.cpp:1753         (rsi.cpp:2208)
 *
 * Conventions: inputs are borrowed for the call; results are owned by the rsi_result and freed
 * by rsi_result_free; every entry point returns RSI_OK or a negative rsi_status and leaves a
 * message for rsi_hot_last_error().  There is NO CPU fallback: without a HIP device
 * rsi_hot_create() fails with RSI_ERR_NO_DEVICE.  One context per host thread / GPU; contexts are
 * independent (no shared mutable globals), so chromosomes can be processed concurrently.
 */
#ifndef RSI_HOT_H
#define RSI_HOT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum rsi_status {
  RSI_OK = 0,
  RSI_ERR_NO_DEVICE = -1,    /* no HIP device / extension unusable: the product never falls back */
  RSI_ERR_BAD_ARG = -2,
  RSI_ERR_HIP = -3,          /* a HIP runtime call failed */
  RSI_ERR_TOO_SMALL = -4,    /* chromosome shorter than 20*202 bases under GC adjust (gccontent.cpp:66-71) */
  RSI_ERR_UNSUPPORTED = -5,  /* e.g. negative depth, a scan longer than 4 M lengths, N-run list overflow */
  RSI_ERR_INTERNAL = -6
} rsi_status;

/* rsi:: statics that the path reads (rsi.h:54-122, defaults rsi.cpp:34-98). */
typedef struct rsi_params {
  int32_t m;          /* -m      bin size, forced odd by the caller as rsi.cpp:2061-2064 does */
  int32_t gcadjust;   /* !-NOGC  */
  int32_t trans;      /* 0 = NBN (-NB, default), 1 = MED (-MED), 2 = ALL (-ALL) */
  int32_t merge;      /* !-nomerge */
  int32_t maxchkbp;   /* -maxchkbp */
  int32_t debug;      /* -debug */
  double cap;         /* -cap */
  double epsilon;     /* -e */
  double threshold;   /* -threshold */
  double chklen;      /* -reflen */
  double minmlen;
  double buffer;
  double p;
} rsi_params;

/* POD mirror of cnv_st (rsi.h:8-51); qscore is the SCORE column of cnv_format1 (rsi.cpp:583-615). */
typedef struct rsi_call {
  int32_t start, end;   /* 0-based indices on the reference (N regions re-inserted), inclusive */
  int32_t type;         /* 0 DEL, 1 DUP, 2 UNKNOWN (rsi.h:4-6) */
  int32_t geno, status, length, qscore, pad;
  double score, p1, cnvmed, cnvsd, cnviqr, refmed, refsd, refiqr;
} rsi_call;

/* Per-chromosome scalars the reference keeps in rsi:: globals or prints to its log. */
typedef struct rsi_chrom_stats {
  int64_t n;            /* chromosome length */
  int64_t n_compact;    /* after N-region removal (rsi::end) */
  int64_t nbins;
  int32_t n_noncode;    /* padded N regions */
  int32_t Lmax;         /* scan length actually used */
  double gc_rdmean;     /* "RD mean before GC adjust" (gccontent.cpp:154) */
  double cap_median;    /* median used by apply_cap (loaddata.cpp:233) */
  double RDmedian, RDsd;
  double nb_mad, nb_r, nb_tmin;
  double tmedian1, tsigma1, tlamda1;   /* first scan pass */
  double tmedian2, tsigma2, tlamda2;   /* second scan pass */
  int32_t trim_escapes; /* trim walks that left the array (the reference aborts there, App. A Q12) */
  int32_t inexact_sums; /* bins whose value breaks the exact-window-sum precondition */
  double t_device_ms;   /* wall time of the call, inputs already on the device */
  double t_kernels_ms;  /* sum of HIP-event times around the per-base kernels */
  int64_t byte_escapes; /* bases of depth >= 255: the per-base kernels fetch those from the int32 array instead of the byte copy */
  int32_t scan_tiles;        /* tiles of 256 bins per scan pass */
  int32_t scan_tiles_listed; /* tiles the detection passes of the (last) scan listed for the exact sweep, both passes together */
} rsi_chrom_stats;

typedef struct rsi_ctx rsi_ctx;
typedef struct rsi_result rsi_result;

/* Reference defaults (rsi.cpp:34-98). */
void rsi_default_params(rsi_params* p);

/* Context on HIP device `device`.  Returns NULL on failure; *status receives the reason. */
rsi_ctx* rsi_hot_create(int device, int* status);
void rsi_hot_destroy(rsi_ctx* ctx);
const char* rsi_hot_last_error(const rsi_ctx* ctx);   /* ctx may be NULL: last global error */

/* One chromosome, inputs in host memory: depth[n] raw per-base depth, fasta[n] sequence bytes. */
int rsi_hot_run(rsi_ctx* ctx, const rsi_params* p, const int32_t* depth, const uint8_t* fasta, int64_t n,
                rsi_result** out);

/* ---- Depth text ingestion on the device --------------------------------------------
 * Replaces the parse loop of load_data_from_text (loaddata.cpp:496-517): "pos depth" lines; empty lines
 * and '#' lines skipped; pos < 1 skipped; reading stops at the first pos >= n (the last base is never
 * set); RD[pos-1] = depth; positions that never appear stay 0.  The file is streamed to HBM in pinned
 * chunks and parsed by a kernel; files whose positions are not strictly increasing (where the order-
 * dependent rules matter) are parsed by the sequential host loop instead (stats->fallback = 1). */
typedef struct rsi_text_stats {
  int64_t bytes, lines, stored, beyond;   /* file size; lines with pos >= 1; lines stored; lines with pos >= n */
  int32_t fallback, pad;                  /* 1: parsed on the host (unsorted positions) */
  double t_total_ms, t_parse_kernel_ms;   /* wall time of the load; summed kernel time when timing is on */
} rsi_text_stats;
/* Parses `path` into the context's device depth buffer (int32[n]); rsi_hot_fetch_i32("depth_in") reads it back. */
int rsi_hot_load_depth_text(rsi_ctx* ctx, const char* path, int64_t n, rsi_text_stats* stats);
/* rsi_hot_load_depth_text + rsi_hot_run on the loaded depth: fasta[n] in host memory.  stats may be NULL. */
int rsi_hot_run_text(rsi_ctx* ctx, const rsi_params* p, const char* depth_path, const uint8_t* fasta, int64_t n,
                     rsi_result** out, rsi_text_stats* stats);

/* ---- BAM pileup -> per-base depth, GPU-assisted -------------------------------------
 * Replaces the read loop of load_data_from_bam (loaddata.cpp:277-333): reads of chromosome `chrom`
 * (all of them, as bam_iter_query(ref, 0, 0x7fffffff) yields) that pass pos != 0, mapq >= minq, not
 * secondary, not duplicate; every base of an M or '=' CIGAR operation with base quality >= min_baseq
 * adds one to the depth at its reference position as resolve_cigar_pos computes it (samfunctions.cpp:
 * 38-100, including that '=' / 'X' do not advance the position).  The host inflates the BGZF blocks
 * (threads) and finds record boundaries; filters, CIGAR and qualities are evaluated on the device, one
 * thread per read, into a difference array that a scan turns into the depth.  A `BAM.bai` next to the
 * file is used to start at the chromosome's first read; without it the file is scanned from the top.
 * Reference defaults: minq 0 (-q), min_baseq 13 (-Q), rsi.cpp:57-58. */
typedef struct rsi_bam_stats {
  int64_t n;                         /* length of the chromosome (header) */
  int64_t bytes_compressed, bytes_inflated, records, on_chrom, used, runs;   /* records walked, reads of `chrom`, reads that passed the filters, runs of counted bases */
  int32_t tid, indexed;              /* reference id of `chrom`; 1 if the .bai was used */
  double t_total_ms, t_inflate_ms, t_walk_ms, t_wait_ms;   /* wall time of the load; of which: inflate (threads), record walk, waiting for the device */
  int64_t malformed;                 /* records whose name / CIGAR / sequence lengths overrun their block_size: skipped, never followed */
} rsi_bam_stats;
/* Depth of `chrom` into the context's device depth buffer (int32[stats->n]); rsi_hot_fetch_i32("depth_in") reads it back. */
int rsi_hot_load_depth_bam(rsi_ctx* ctx, const char* bam_path, const char* chrom, int minq, int min_baseq, rsi_bam_stats* stats);
/* RP / Q0 annotation of the final calls (cnv_stat, pairrd.cpp:622-748; host only): the insert-size statistics are
 * sampled from the window the reference uses (1 Mb of reads from position 10 Mb of the chromosome on, pairrd.cpp:636),
 * then, per call, the read pairs within max(1000, length) <= 5000 bases are classified.  Needs BAM.bai.  The result's
 * rows (rsi_result_format_row) print the values afterwards; rsi_result_pairs reads them (-1 / -1.0 before annotation,
 * rsi.h:49-50).  On chromosomes shorter than 10 Mb the reference crashes in this step; here the statistics keep their
 * defaults (-1) there. */
int rsi_result_annotate_bam(rsi_result* r, const char* bam_path, const char* chrom);
int rsi_result_pairs(const rsi_result* r, int i, int32_t* rp, double* q0);
/* Fixed-layout summary of a chromosome's result for the one exchange of a multi-GPU run (the gather of per-chromosome
 * results that replaces the reference's sequential output loop, rsi.cpp:1594-1608):
 *   out[0..7]  = chromosome id (the caller's), chromosome median, SD, number of final calls, number stored here, 0, 0, 0
 *   then RSI_SUMMARY_CALL doubles per stored call: start, end, type, qscore, cnvmed, cnviqr, refmed, refiqr,
 * at most max_calls of them (stored < number of calls means the block was too small: callers treat that as an error).
 * Returns the number of doubles written.  rsi_summary_format_row: the output row of stored call i of such a block. */
#define RSI_SUMMARY_HEAD 8
#define RSI_SUMMARY_CALL 8
int rsi_result_summary(const rsi_result* r, int chrom_id, double* out, int max_calls);
int rsi_summary_format_row(const double* block, int i, const char* chrom, char* buf, int cap);
/* every stored call of the block, one row per line (each ended by '\n'); returns the bytes written, < 0 if buf is too small */
int rsi_summary_format_rows(const double* block, const char* chrom, char* buf, int cap);
/* The diagnostic lines of the chromosome's (last) scan as the reference writes them to its log, in its order: the NB
 * transform's "RD median : " / "RD median absolute deviation : " (rsi.cpp:1140-1141), the first pass' per-L lines
 * (rsi.cpp:1221-1224 "DEL-", 1251-1254 "DUP+": L, bins marked so far, bins, portion), filterstatus' level table (level, bins,
 * float mean; then the two chosen levels; rsi.cpp:991-1002), the second pass' per-L lines.  This is the log sink of SURVEY
 * 8(b) as a pull interface: line i (0-based) into buf; returns i + 1, or 0 when there is no such line. */
int rsi_result_log_line(const rsi_result* r, int i, char* buf, int cap);
/* Reference sequences of the BAM header: names as one '\n'-separated string into names[names_cap], lengths into
 * lengths[max_refs]; returns their number (also when the buffers are too small or NULL), < 0 on error. */
int rsi_bam_references(const char* bam_path, char* names, int names_cap, int64_t* lengths, int max_refs);
/* rsi_hot_load_depth_bam + rsi_hot_run on the loaded depth; fasta[n] in host memory, n must equal the header's length. */
int rsi_hot_run_bam(rsi_ctx* ctx, const rsi_params* p, const char* bam_path, const char* chrom, int minq, int min_baseq,
                    const uint8_t* fasta, int64_t n, rsi_result** out, rsi_bam_stats* stats);
/* Same, inputs already resident in device memory (HBM); they are not modified. */
int rsi_hot_run_device(rsi_ctx* ctx, const rsi_params* p, const void* d_depth, const void* d_fasta, int64_t n,
                       rsi_result** out);

/* Results.  which: 0 = calls after sd_filters (what write_cnv_to_file prints),
 *                  1 = detectcnv output before sd_filters,
 *                  2 = bin-space segments after the scan (rsicnvnbn / rsicnvmed output),
 *                  3 = bin-space segments after areblockscnv + sort. */
int rsi_result_ncalls(const rsi_result* r, int which);
const rsi_call* rsi_result_calls(const rsi_result* r, int which);
const rsi_chrom_stats* rsi_result_stats(const rsi_result* r);
/* Padded N regions as (start,end) inclusive pairs; returns the number of regions. */
int rsi_result_noncode(const rsi_result* r, int32_t* pairs, int cap);
/* One output row exactly as cnv_format1 prints it (rsi.cpp:581-631), without the newline. */
int rsi_result_format_row(const rsi_result* r, int i, const char* chrom, char* buf, int cap);
void rsi_result_free(rsi_result* r);

/* Intermediates for the parity tests (copied device -> host on demand while the context still
 * holds the chromosome: valid until the next rsi_hot_run* on the same context).
 * int32 names: "rd_gc" (after GC adjust, n), "rd_concat" (capped + compacted, n_compact),
 *              "binmedint", "status1", "status1f", "status2" (nbins each)
 * f32   names: "binnb", "binmed" (nbins)
 * i64   names: "binsum" (nbins)
 * Returns the element count (also when out == NULL), or a negative rsi_status. */
int64_t rsi_hot_fetch_i32(rsi_ctx* ctx, const char* name, int32_t* out, int64_t cap);
int64_t rsi_hot_fetch_f32(rsi_ctx* ctx, const char* name, float* out, int64_t cap);
int64_t rsi_hot_fetch_i64(rsi_ctx* ctx, const char* name, int64_t* out, int64_t cap);

/* Test hook: filterstatus' per-level sums (rsi.cpp:967-976: float accumulation in index order per status level) of HOST arrays
 * T[nb], status[nb] (values in [-Lmax, Lmax]) through the device's exact parallel form; sums / counts: 2 Lmax + 1 entries, index
 * = level + Lmax.  counts[Lmax] == -1: the device declined the input (the pipeline then runs the sequential loop itself). */
int rsi_hot_debug_level_sums(rsi_ctx* ctx, const float* T, const int32_t* status, int64_t nb, int Lmax, float* sums, int32_t* counts);
/* Test hook: one scan pass (rsistatus) over host arrays with the caller's thresholds; status[nb] out, info[4] = tiles the
 * detection pass listed, trimming walks that left the array, inexact-threshold flags, 0. */
int rsi_hot_debug_scan(rsi_ctx* ctx, const float* T, const int32_t* medint, int64_t nb, double RDmedian, double tmedian, double tlamda,
                       int Lmax, int32_t* status, int32_t* info);

/* Timing hooks for bench.py: per-kernel HIP-event times (ms) of the last run, by kernel name.
 * names/ms receive up to cap entries; returns the number of timed launches. */
int rsi_hot_kernel_times(const rsi_ctx* ctx, const char** names, float* ms, int cap);
/* Host wall-clock per pipeline phase of the last run (ms, includes waits on the device). */
int rsi_hot_phase_times(const rsi_ctx* ctx, const char** names, double* ms, int cap);
/* Per-kernel event timing: 0 off, 1 an event pair around every launch, 2 around the per-base (HBM-bound)
 * kernels only -- some sixty launches per chromosome make the event records themselves cost 13 % of a pooled step. */
void rsi_hot_set_timing(rsi_ctx* ctx, int on);
/* Mode 3: an event pair around the launches of ONE kernel, named as in rsi_hot_kernel_times' table (e.g. "gc_hist",
 * "cap_compact_bin", "rsi_scan"): what bench.py's timed steps use for the kernel that an untimed pass measured as the
 * dominant one.  NULL / "" = "cap_compact_bin". */
void rsi_hot_set_timing_kernel(rsi_ctx* ctx, const char* name);

/* ---- Pool: several chromosomes in flight on one GPU ------------------------------------------
 * The reference's per-chromosome loop (rsi.cpp:2189-2217) has independent iterations.  A pool owns
 * `nworkers` host threads (created with the pool, asleep while no run is queued; worker 0 is a thread waiting for a run),
 * each with its own context (stream + workspace); the chromosomes of a run are handed out longest first.  Runs may be queued
 * (rsi_pool_submit / rsi_pool_wait below) and are worked on in submission order; several pools (one per GPU) are independent.  At most three HBM-bound
 * per-base phases are in flight per GPU (rsi_pool_set_schedule); bin-level and candidate kernels,
 * copies and the host stages of different chromosomes overlap. */
typedef struct rsi_pool rsi_pool;
#define RSI_MAX_TIMED 64
typedef struct rsi_batch_times {   /* accumulated over rsi_pool_run calls; zero it to start */
  int32_t nkernels, nphases;
  const char* kernel_name[RSI_MAX_TIMED];
  double kernel_ms[RSI_MAX_TIMED];       /* sum of HIP-event durations */
  int64_t kernel_launches[RSI_MAX_TIMED];
  int64_t kernel_bases[RSI_MAX_TIMED];   /* sum over launches of the chromosome length */
  const char* phase_name[RSI_MAX_TIMED];
  double phase_ms[RSI_MAX_TIMED];        /* host wall-clock per pipeline phase, summed over workers */
} rsi_batch_times;

rsi_pool* rsi_pool_create(int device, int nworkers, int* status);
void rsi_pool_destroy(rsi_pool* pool);
int rsi_pool_workers(const rsi_pool* pool);
rsi_ctx* rsi_pool_worker(rsi_pool* pool, int w);
void rsi_pool_set_timing(rsi_pool* pool, int on);
void rsi_pool_set_timing_kernel(rsi_pool* pool, const char* name);   /* rsi_hot_set_timing_kernel on every worker */
/* Scheduling of the pool's workers on the GPU.  isolate != 0: a chromosome's per-base (HBM-bound) phase runs
 * alone on the chip -- bin-level work of the other workers waits -- so that every streaming launch is a clean
 * bandwidth sample (profiling); 0 (default): bin-level work overlaps it (about 20 % more throughput).
 * streamers: per-base phases allowed in flight at once (default 2; ignored while isolating). */
void rsi_pool_set_schedule(rsi_pool* pool, int isolate, int streamers);
const char* rsi_pool_last_error(const rsi_pool* pool);
/* d_depth[i], d_fasta[i], n[i]: chromosome i, resident in HBM.  out[i] receives its result (or NULL),
 * status[i] (optional) its rsi_status; returns the first failure or RSI_OK.  times may be NULL. */
int rsi_pool_run(rsi_pool* pool, const rsi_params* p, int nchrom, const void* const* d_depth,
                 const void* const* d_fasta, const int64_t* n, rsi_result** out, int* status,
                 rsi_batch_times* times);

/* ---- Plot files (plotcnv.cpp:245-610; host only) ---------------------------------------------------------
 * The gnuplot data (.dat) and script (.gp) file of one call as plot_icnv writes them.  rd: the capped, GC-adjusted per-base
 * depth with the removed N regions back in as zeros (expand_data, loaddata.cpp:140; rsi_plot_expand builds it from
 * rsi_hot_fetch_i32("rd_concat") and rsi_result_noncode); chrom_median: _median of that array (plot::RDmed).  The reference
 * pipes the script through gnuplot and deletes both files; the writer stops at the files. */
int rsi_plot_expand(const int32_t* rdc, int64_t ncompact, const int32_t* regions, int npairs, int32_t* out, int64_t n);
int rsi_plot_write_files(const rsi_call* c, const char* title, const int32_t* rd, int64_t n, double chrom_median, int m,
                         double minmlen, double chklen, const char* format, double gnuplot_version, const char* datfile,
                         const char* gpfile, const char* imgfile);

/* The same from host memory (load_data_from_text / load_data_from_bam leave the reference's RD in host memory, loaddata.cpp:473-539):
 * every worker moves its chromosome over its own stream (pinned buffers: asynchronous DMA; pageable ones work, staged by
 * the runtime), so the transfers of some chromosomes overlap the kernels of others.  PCIe-bound at 5 bytes per base. */
int rsi_pool_run_host(rsi_pool* pool, const rsi_params* p, int nchrom, const int32_t* const* depth,
                      const uint8_t* const* fasta, const int64_t* n, rsi_result** out, int* status,
                      rsi_batch_times* times);

/* Samples back to back (the reference is started once per sample; a sequencing centre runs them one after the other): queue a
 * run and return at once.  The pool's workers take chromosomes in submission order, so a worker that finds no chromosome left
 * in one run starts on the next -- the last, short chromosomes of one sample run beside the first, long ones of the next
 * instead of leaving most of the GPU idle.  The pointer arrays are copied; `out`, `status`, `times` and the data they point
 * to must stay valid until rsi_pool_wait has returned for the ticket.  Returns 0 for bad arguments.  rsi_pool_wait returns
 * the run's worst status; the calling thread works as one of the pool's workers while it waits (on the runs up to its own).
 * rsi_pool_run is submit + wait; any number of threads may call the three.  Every ticket must be waited for before
 * rsi_pool_destroy. */
uint64_t rsi_pool_submit(rsi_pool* pool, const rsi_params* p, int nchrom, const void* const* d_depth,
                         const void* const* d_fasta, const int64_t* n, rsi_result** out, int* status,
                         rsi_batch_times* times);
int rsi_pool_wait(rsi_pool* pool, uint64_t ticket);

#ifdef __cplusplus
}
#endif
#endif
