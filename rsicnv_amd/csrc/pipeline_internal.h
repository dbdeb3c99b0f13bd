// pipeline_internal.h -- what the translation units behind include/rsi_hot.h share: the context (device workspace,
// stream, pinned mailbox), the pool's gate, the small-transfer and timing helpers, the layout of the `small` buffer.
// pipeline.hip: one chromosome through the path; ingest.hip: depth from text / BAM; pool.hip: several chromosomes at once.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/prctl.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/rsi_hot.h"
#include "bam_host.h"
#include "host_calls.h"
#include "hostmath.h"
#include "kernels.h"

using namespace rsik;
using rsih::Candidate;
using rsih::Region;


#include "gate.h"   // GpuGate: who may run a per-base phase when (plain C++, also built under ThreadSanitizer: tests/sanitize)

namespace rsip {


inline std::string g_last_error;   // last failure of any context (diagnostic; guarded by g_err_mu)
inline std::mutex g_err_mu;
inline void set_global_error(const std::string& m) { std::lock_guard<std::mutex> lk(g_err_mu); g_last_error = m; }

// Growth hint of the running thread: (largest chromosome its pool has been handed) / (this one).
// A buffer that has to grow is sized for the largest chromosome right away, so that after a
// worker's first chromosome no hipFree/hipMalloc -- a device-wide synchronisation that stalls every
// other worker, and slow on recycled VRAM -- happens in steady state.
inline thread_local double tl_grow = 1.0;
inline thread_local double tl_grow_ms = 0.0;   // time this thread spent re-allocating during the current run

// RSI_HOT_POISON=1 (debugging): every new device and pinned allocation is filled with 0xA5 instead of whatever the driver
// hands out (zero pages in a fresh process, another context's leftovers in a long one)
inline bool poison_allocations() { static const bool on = getenv("RSI_HOT_POISON") && atoi(getenv("RSI_HOT_POISON")) != 0; return on; }

struct DevBuf {   // grow-only device allocation
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    const auto t0 = std::chrono::steady_clock::now();
    const bool regrow = p != nullptr;   // a buffer whose size depends on the data turned out too small: be generous this time
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t scaled = (size_t)((double)bytes * tl_grow);
    const size_t want = (regrow ? 2 * scaled : scaled + scaled / 8) + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    if (e == hipSuccess && poison_allocations()) {   // RSI_HOT_POISON=1: a read of memory nobody wrote shows up at once
      e = hipMemset(p, 0xA5, want);
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);   // (the fill runs on the null stream: done before the context's own stream uses the buffer)
    }
    tl_grow_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return e;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
  ~DevBuf() { if (p) (void)hipFree(p); }
};

struct PinBuf {   // grow-only pinned host allocation: destination of the large device -> host copies.  From pageable memory
                  // the runtime stages such a copy synchronously under its own lock, which stalls every other worker's
                  // HIP calls for milliseconds per copy.
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t scaled = (size_t)((double)bytes * tl_grow);
    const size_t want = scaled + scaled / 8 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    if (e == hipSuccess && poison_allocations()) memset(p, 0xA5, want);
    return e;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
  ~PinBuf() { if (p) (void)hipHostFree(p); }
};

struct KernelTime { const char* name; hipEvent_t a, b; };

inline double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}


}  // namespace rsip

using rsip::DevBuf;
using rsip::PinBuf;
using rsip::KernelTime;
using rsip::GpuGate;

struct rsi_result {
  std::vector<rsi_call> lists[4];
  std::vector<int32_t> noncode;   // pairs
  std::vector<int32_t> rp;        // per final call, after rsi_result_annotate_bam
  std::vector<double> q0;
  // per-L counts of newly marked bins of the four sweeps of the (last) scan -- pass 1 DEL, DUP, pass 2 DEL, DUP -- and the L
  // each sweep stopped at: the reference's "DEL-" / "DUP+" log lines (rsi.cpp:1221-1224, 1251-1254)
  std::vector<uint32_t> level_log[4];
  uint32_t stop_levels[4] = {0, 0, 0, 0};
  std::vector<std::string> fs_lines;   // filterstatus' level table of that scan, as logged (rsi.cpp:991-1002)
  bool log_nb = false;                 // that scan ran on the NB transform: its two log lines come first (rsi.cpp:1140-1141)
  rsi_chrom_stats stats;
  rsi_params params;
};

struct rsi_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t sync_ev = nullptr;   // event every wait on the stream polls (stream_wait)
  hipStream_t copy_stream = nullptr;   // the bin medians' copy to the host (block tests) runs beside the transform and scan kernels
  hipEvent_t copy_ev = nullptr;
  bool copy_pending = false, copy_recorded = false;   // a copy is queued on copy_stream / copy_ev was recorded behind it
  std::string err;
  int timing = 0;   // 0 off, 1 HIP events around every launch, 2 around the per-base kernels only, 3 around the kernel named timing_kernel only
  std::string timing_kernel = "cap_compact_bin";
  std::vector<KernelTime> ktimes;
  std::vector<hipEvent_t> event_pool;
  size_t event_next = 0;
  // workspace
  DevBuf in_depth, in_fasta;                 // staging for the host-pointer entry point
  DevBuf in_d8, in_esc;                      // ... its narrowed form: the depth as bytes, the values that did not fit (pos | val)
  PinBuf h_d8;                               // pinned: the bytes and the list on their way to the device
  DevBuf text_dev[2], text_wg;               // depth text ingestion: two chunks of file bytes in HBM, per-workgroup order records
  char* text_pin[2] = {nullptr, nullptr};    // pinned staging for the file bytes
  size_t text_pin_cap = 0;
  int64_t n_in = 0;                          // length of the depth currently in in_depth
  DevBuf gcbits, nbits, rd_gc, rdc, binmed, binsum, tnb, tmed, first_del;
  DevBuf depth8, rescaled8;   // byte copies: the raw depth (K2 writes it, K3' streams it), the rescaled depth (K3' -> K4')
  DevBuf slabs;   // per-workgroup partial results of the streaming kernels
  DevBuf gsum;    // group sums of the in-kernel slab folds (device_util.h)
  DevBuf status1, status1f, status2, hist_val, hist_res, hist_f, small, thr, runs, run_se, scratch, items, best;
  DevBuf cand_jobs, cand_chains, cand_outs, cand_i32, cand_i64, cand_mid, cand_hist;   // candidate tests on the device (kernels_cand.hip)
  DevBuf sharpen_ws;          // workspace of k_sharpen_edges, cleared when (re)allocated
  DevBuf fs_ws, fs_out;       // filterstatus' level sums on the device (kernels_fs.hip)
  DevBuf scan_tiles;          // tiles the scan's detection pass lists for the exact sweep
  DevBuf scan_ws;             // the exact sweep's tiles in device memory, for scans too long for LDS (scan_tile_workspace_bytes)
  DevBuf joint_tot;           // K2j's folded joint histogram [GC count][depth byte] + escapes
  // K4j queued behind K2j without a host round trip needs its launch configuration before the cap is known: the cap of the
  // context's previous chromosome under the same flags (one sample: one depth), checked on the device and again by the host
  int32_t spec_capval = -1; int32_t spec_m = 0; double spec_cap = 0.0; bool spec_gc = true;   // ... and whether that chromosome ran with the GC adjustment (the queued K4 of a -NOGC run takes the raw bytes)
  int sharpen_ws_jobs = 0;    // jobs it is laid out for
  // host mirrors kept for rsi_hot_fetch_* (what the last run left on the device)
  int64_t n = 0, ncompact = 0, nb = 0;
  bool have_gc = false, have_nb = false, have_med = false;
  DevBuf rdc8;                               // the capped, compacted depth as bytes (what K4' writes; the candidate kernels read it in place)
  bool rdc_is_bytes = false, rdc_valid = false;   // rdc8 is the last run's array / the int32 rdc holds it too (materialize_rdc)
  bool rd_gc_valid = false;                  // rd_gc holds the rescaled depth of the last run (else rsi_hot_fetch builds it on demand)
  const int32_t* last_depth = nullptr;       // device input of the last run (borrowed; needed to build rd_gc on demand)
  int last_scan_med = 0;
  // wall-clock per pipeline phase of the last run (host view, includes waits), for bench.py
  std::vector<std::pair<const char*, double>> phases;
  // when the context belongs to a pool: arbitration of the GPU between workers
  DevBuf scan_work_big;   // work blocks of a scan longer than kMaxL (scan_pass)
  PinBuf h_T, h_status, h_status2, h_medint;   // pinned host copies of the bin arrays (filterstatus, block tests)
  // pinned host mailbox: small transfers in both directions go through it (see copy_d2h / copy_h2d)
  char* mailbox = nullptr;
  size_t mb_cap = 0, mb_used = 0;
  struct Pending { void* dst; const void* src; size_t bytes; };
  std::vector<Pending> pending;   // mailbox -> destination copies to finish at the next synchronisation
  int32_t* mirror = nullptr;  // pinned host mirror of the compacted depth (DepthPager), grow-only
  size_t mirror_cap = 0;
  GpuGate* gate = nullptr;
  std::atomic<int64_t> reserve_n{0};   // largest chromosome the pool has seen: workspace growth is sized for it (written at submission, read when a chromosome starts)
  bool poisoned = false;      // a wait hit its deadline with work still queued (ctx_sync): no further runs
  bool gate_shared = false;   // RSI_HOT_ISOLATE_STREAMING=1: bin-level kernels wait while a per-base phase runs (clean kernel timings, ~20 % less throughput)
};

namespace rsip {


#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      ctx->err = std::string(#expr) + ": " + hipGetErrorString(e__);                          \
      set_global_error(ctx->err);                                                             \
      return RSI_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

// Wait for the context's stream without burning a core: poll briefly (most waits are a few microseconds), then nap
// between polls.  With one busy-spinning thread per worker a 16-CPU quota is exhausted by the waits alone and the whole
// process gets throttled; a blocking hipEventSynchronize sleeps until the completion interrupt, and when nothing else
// is running on the GPU (one chromosome alone in a pool) that wake-up now and then came 30 ms late on some boxes.
// The naps need a small timer slack (10 us, not 10 + 50): lowered for the duration of the wait only -- worker 0 of a pool
// runs on the caller's thread, whose settings are the host application's.  A queue that makes no progress for a minute is
// reported as a failure instead of being polled for ever.
constexpr double kStreamWaitDeadlineMs = 60000.0;
// `busy_pool`: many chromosomes are in flight on this GPU, the wait will be long and other workers need the cores -- a short
// spin, then naps (sixteen workers spinning two thousand queries per wait kept twelve cores busy for nothing: the step is
// the same with none, and a pool of 24 workers on a 16-core box went from 14 to 25 ms with them).  A lone chromosome's waits
// are short and nobody else wants the core: spin.  RSI_HOT_SPIN overrides the count.
inline hipError_t event_wait(hipEvent_t ev, bool busy_pool = false) {   // an event that has been recorded
  static const int spin_env = [] { const char* v = getenv("RSI_HOT_SPIN"); return v ? atoi(v) : -1; }();
  const int spins = spin_env >= 0 ? spin_env : (busy_pool ? 50 : 2000);
  hipError_t e = hipSuccess;
  for (int spin = 0; spin < spins; ++spin) {
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
  }
  const int slack_before = prctl(PR_GET_TIMERSLACK, 0UL, 0UL, 0UL, 0UL);
  (void)prctl(PR_SET_TIMERSLACK, 2000UL, 0UL, 0UL, 0UL);
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    struct timespec nap = {0, 10000};   // 10 us
    nanosleep(&nap, nullptr);
    e = hipEventQuery(ev);
    if (e != hipErrorNotReady) break;
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > kStreamWaitDeadlineMs) { e = hipErrorLaunchTimeOut; break; }
  }
  if (slack_before > 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)slack_before, 0UL, 0UL, 0UL);
  return e;
}
inline hipError_t stream_wait(hipStream_t stream, hipEvent_t ev, bool busy_pool = false) {
  const hipError_t e = hipEventRecord(ev, stream);
  return e != hipSuccess ? e : event_wait(ev, busy_pool);
}

// Small transfers go through a pinned mailbox.  A hipMemcpyAsync on pageable memory is staged by the
// runtime and blocks the calling thread; with a dozen workers issuing some fifty small copies per
// chromosome that serialises them.  From / to pinned memory the copy is a plain asynchronous DMA:
//  * host -> device: the bytes are parked in the mailbox first, so the caller's buffer is free at once;
//  * device -> host: the bytes land in the mailbox and are moved to their destination by the next
//    ctx_sync() -- exactly when the caller may look at them anyway.
// Slots live until the end of the run (bump allocation); large transfers take the direct path.
constexpr size_t kMailboxBytes = size_t(8) << 20, kMailboxMaxCopy = size_t(512) << 10;
inline void* mb_alloc(rsi_ctx* ctx, size_t bytes) {
  if (!ctx->mailbox) {
    // mapped + coherent: kernels write small results straight into it (device_util.h, export_words), visible to the host
    // once the stream has passed the kernel
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->mailbox), kMailboxBytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { ctx->mailbox = nullptr; return nullptr; }
    void* dev_view = nullptr;
    if (hipHostGetDevicePointer(&dev_view, ctx->mailbox, 0) != hipSuccess || dev_view != (void*)ctx->mailbox) {   // unified addressing: same pointer
      (void)hipHostFree(ctx->mailbox); ctx->mailbox = nullptr; return nullptr;
    }
    if (poison_allocations()) memset(ctx->mailbox, 0xA5, kMailboxBytes);
    ctx->mb_cap = kMailboxBytes;
  }
  const size_t need = (bytes + 63) & ~size_t(63);
  if (ctx->mb_used + need > ctx->mb_cap) return nullptr;
  void* p = ctx->mailbox + ctx->mb_used;
  ctx->mb_used += need;
  return p;
}
inline hipError_t copy_d2h(rsi_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  void* slot = bytes <= kMailboxMaxCopy ? mb_alloc(ctx, bytes) : nullptr;
  if (!slot) return hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
  ctx->pending.push_back({dst, slot, bytes});
  return hipMemcpyAsync(slot, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
}
inline hipError_t copy_h2d(rsi_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  void* slot = bytes <= kMailboxMaxCopy ? mb_alloc(ctx, bytes) : nullptr;
  if (!slot) return hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
  memcpy(slot, src, bytes);
  return hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream);
}
// Every wait also collects the launch errors of the chain in front of it: a rejected launch (too much dynamic LDS, a bad
// grid) raises no error at the launch site -- the launch wrappers return nothing -- and the copies and event calls behind it
// still succeed; without this check the run would return RSI_OK on stale data.
inline hipError_t ctx_sync(rsi_ctx* ctx) {
  if (ctx->poisoned) return hipErrorLaunchTimeOut;
  hipError_t e = stream_wait(ctx->stream, ctx->sync_ev, ctx->gate != nullptr && !ctx->gate->lonely());
  // A queue that outlived the wait's deadline is still running: its kernels write the workspaces and the mapped mailbox of
  // this context, so the context takes no further run (a next run would reset the mailbox and reuse the buffers under
  // them) -- every entry point refuses a poisoned context; rsi_hot_destroy is what is left to do with it.
  if (e == hipErrorLaunchTimeOut) ctx->poisoned = true;
  const hipError_t launch = take_launch_error();   // first failed launch of this thread since the last wait (kernels.h, RSI_LAUNCH)
  if (e == hipSuccess && launch != hipSuccess) e = launch;
  if (e == hipSuccess) for (const rsi_ctx::Pending& c : ctx->pending) memcpy(c.dst, c.src, c.bytes);
  ctx->pending.clear();
  return e;
}
// The bin medians' copy to the host runs on the device's shared copy stream (pipeline.hip, bin_level_stages).  Its join has the
// deadline and the consequences of every other wait: a copy that does not come back within it poisons the context (the DMA may
// still write h_medint).  A copy that was queued but whose event could not be recorded is waited for on the stream itself.
inline hipError_t join_copy(rsi_ctx* ctx) {
  if (!ctx->copy_pending) return hipSuccess;
  ctx->copy_pending = false;
  hipError_t e = ctx->copy_recorded ? event_wait(ctx->copy_ev, ctx->gate != nullptr && !ctx->gate->lonely()) : hipStreamSynchronize(ctx->copy_stream);
  ctx->copy_recorded = false;
  if (e == hipErrorLaunchTimeOut) ctx->poisoned = true;
  return e;
}
inline void mailbox_reset(rsi_ctx* ctx) { ctx->mb_used = 0; ctx->pending.clear(); }
// First thing an entry point does with a context: refuse a poisoned one, forget errors that are not this run's (the calling
// thread is the host application's: torch or RCCL may have left a benign failure in the runtime's per-thread slot).
inline bool ctx_enter(rsi_ctx* ctx) {
  if (ctx->poisoned) {
    ctx->err = "context unusable: an earlier wait gave up after 60 s with kernels still queued (destroy it)";
    set_global_error(ctx->err);
    return false;
  }
  drain_stale_errors();
  return true;
}
#define CTX_SYNC() ctx_sync(ctx)

inline int fail(rsi_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  set_global_error(msg);
  return code;
}

// offsets into the `small` buffer (accumulators and little lists), all 256-byte aligned
constexpr size_t kOffGcAcc = 0;                                   // GcAccum
constexpr size_t kOffJointInfo = 3328;                            // JointInfo (behind GcAccum's 3256 bytes)
constexpr size_t kOffPhase = 3344;                                // PhaseParams (32 bytes)
constexpr size_t kOffValAux = 4096;                               // ValueHistAux
constexpr size_t kOffMinMax = 4608;                               // MinMaxF
constexpr size_t kOffCounters = 4864;                             // uint32[8]: scan counters, list counts
constexpr size_t kOffRawMin = 5120;                               // uint32
constexpr size_t kOffValMedian = 5184;                            // ValueMedian (24 bytes)
constexpr size_t kOffDone = 5376;                                 // arrival counters of the in-kernel folds: kDoneStride uint32 per kernel
constexpr size_t kDoneStride = 48;                                // >= kFoldGroups + 1
constexpr size_t kDoneBinSlot = 4 * kDoneStride;                     // the bin-level kernels' counters: K4's slot (2) is two strides wide (kFoldGroupsAdd + 1 counters)
static_assert(kOffDone + (kDoneBinSlot + 16) * 4 <= 6400 && kDoneStride >= 32 + 1 && 2 * kDoneStride >= 64 + 2, "arrival counters inside the header (device_util.h: kFoldGroups, kFoldGroupsAdd)");
constexpr size_t kHeaderBytes = 6400;                             // everything above: cleared by the first kernel of a run (K1), handed to the host as one block
constexpr uint32_t kMaxTransitions = 1u << 16;
constexpr size_t kOffNtrans = kHeaderBytes;                       // uint64[kMaxTransitions] N-run boundaries, right behind the header:
                                                                  // the header and the first entries travel as one transfer
constexpr size_t kOffTable = kOffNtrans + (size_t)kMaxTransitions * 8;   // double[202]
constexpr size_t kOffGrid = kOffTable + 1792;                     // GridMedian[2]: the two links of a median -> MAD chain
constexpr size_t kOffScanPass = kOffGrid + 1024;                  // 2 x ScanPassWork (one per rsistatus pass)
constexpr size_t kScanPassBytes = kScanWorkBytes;                 // ScanPassOut, level histograms of the two sweeps (kernels.h)
constexpr size_t kOffBreaks = kOffScanPass + 2 * kScanPassBytes + 128;   // int64 cbreak[4100], cum[4097]
constexpr size_t kSmallBytes = kOffBreaks + 2 * 4100 * 8;
constexpr int kMaxRegions = 4096;
constexpr uint32_t kMaxRunEntries = 1u << 20;
constexpr int kMaxL = kMaxScanL;
constexpr int kHardMaxL = 1 << 22;   // the long form of the scan (pipeline.hip, run_scan) up to here

struct Timer {   // optional HIP-event bracket around one launch
  rsi_ctx* ctx; const char* name; hipEvent_t a = nullptr, b = nullptr;
  Timer(rsi_ctx* c, const char* nm, bool per_base = false) : ctx(c), name(nm) {
    if (!ctx->timing || (ctx->timing == 2 && !per_base) || (ctx->timing == 3 && ctx->timing_kernel != nm)) return;
    auto get = [&]() { if (ctx->event_next == ctx->event_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ctx->event_pool.push_back(e); } return ctx->event_pool[ctx->event_next++]; };
    a = get(); b = get();
    (void)hipEventRecord(a, ctx->stream);
  }
  ~Timer() { if (a) { (void)hipEventRecord(b, ctx->stream); ctx->ktimes.push_back({name, a, b}); } }
};

// Device-built integer histogram -> rsih::Quantiles
inline bool int_quantiles(const std::vector<uint64_t>& h, uint64_t total, rsih::Quantiles& q) {
  return rsih::hist_quantiles_int(h.data(), h.size(), total, q);
}

struct GateShared {   // RAII: a bin-level GPU section of a pooled context (only when the pool isolates streaming)
  GpuGate* g;
  explicit GateShared(rsi_ctx* ctx) : g(ctx->gate_shared ? ctx->gate : nullptr) { if (g) g->lock_shared(); }
  void release() { if (g) { g->unlock_shared(); g = nullptr; } }
  ~GateShared() { release(); }
};

struct Phase {   // wall-clock bracket of one pipeline phase (host view)
  rsi_ctx* ctx; const char* name; double t0; bool open = true;
  Phase(rsi_ctx* c, const char* nm) : ctx(c), name(nm), t0(now_ms()) {}
  void stop() {
    if (!open) return;
    open = false;
    const double dt = now_ms() - t0;
    for (auto& p : ctx->phases) if (p.first == name) { p.second += dt; return; }
    ctx->phases.push_back({name, dt});
  }
  ~Phase() { stop(); }
};

}  // namespace rsip
