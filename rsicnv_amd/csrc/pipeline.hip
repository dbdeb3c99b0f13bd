// pipeline.hip -- the per-chromosome pipeline behind include/rsi_hot.h: device workspace, kernel
// sequencing with the global-reduction barriers of the path (GC table -> cap median -> chromosome
// median/MAD -> NB minimum -> scan thresholds), the small host decisions between kernels, and the
// C ABI.  There is no CPU fallback anywhere in this file: every array-sized computation is a
// kernel from kernels_base.hip / kernels_bin.hip / kernels_cand.hip; the host only walks
// device-built histograms and runs the candidate list logic (host_calls.cpp).
#include <new>
#include "pipeline_internal.h"

using namespace rsik;
using namespace rsip;
using rsih::Candidate;
using rsih::Region;

namespace {

struct ScanPassOut { uint32_t escapes, inexact, ldel, ldup; };   // what one rsistatus pass leaves in `small`

struct ScanOut {
  double tmedian1 = 0, tsigma1 = 0, tlamda1 = 0, tmedian2 = 0, tsigma2 = 0, tlamda2 = 0;
  int Lmax = 0;
  rsih::IntSpan status2;            // status after pass 2: the values inside the runs (run_status / run_ranges), or the context's pinned copy of the whole array
  std::vector<int> run_status;
  std::vector<rsih::IntSpan::Range> run_ranges;
  std::vector<Candidate> segs;
  uint32_t escapes = 0, inexact = 0;
  uint32_t tiles_listed = 0;        // tiles the two detection passes listed for the exact sweep
  // per-L counts of newly marked bins of the four sweeps (pass 1 DEL, DUP, pass 2 DEL, DUP) and the L each stopped at:
  // what the reference logs per L (rsi.cpp:1221-1224, 1251-1254)
  std::vector<uint32_t> level_log[4];
  uint32_t stop_levels[4] = {0, 0, 0, 0};
  std::vector<std::string> fs_lines;   // filterstatus' level table as the reference logs it (rsi.cpp:991-1002)
};

// partition_stat_tp's early return (wufunctions.cpp:371-381): when the selection spans less than the grid step its "median"
// is its MEAN, accumulated in double in index order.  Rare (a chromosome whose bins all carry the same value), so it simply
// runs on the host over a copy of the array.
int selection_mean(rsi_ctx* ctx, const float* d_x, const int32_t* d_mask, int64_t nb, int use_abs, double center, double* mean,
                   uint64_t* count) {
  HIPCHK(ctx->h_T.ensure((size_t)nb * 4));
  HIPCHK(hipMemcpyAsync(ctx->h_T.p, d_x, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (d_mask) {
    HIPCHK(ctx->h_status.ensure((size_t)nb * 4));
    HIPCHK(hipMemcpyAsync(ctx->h_status.p, d_mask, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(CTX_SYNC());
  const float* x = ctx->h_T.as<float>();
  const int* mk = d_mask ? ctx->h_status.as<int>() : nullptr;
  double acc = 0.0;
  uint64_t k = 0;
  for (int64_t i = 0; i < nb; ++i) {
    if (mk && mk[i] != 0) continue;
    const float v = use_abs ? (float)fabs((double)x[i] - center) : x[i];   // RDtmp[i] = abs(RDtrans[i]-tmedian), rsi.cpp:1276
    acc += v;
    ++k;
  }
  *count = k;
  *mean = k ? acc / (double)k : 0.0;
  return RSI_OK;
}

// 0.01-grid median of the selected values of a device float array (partition_stat_tp semantics)
int grid_median(rsi_ctx* ctx, const float* d_x, const int32_t* d_mask, int64_t nb, int use_abs, double center,
                double* med, uint64_t* count) {
  uint8_t* small = ctx->small.as<uint8_t>();
  { Timer t(ctx, "minmax_f32"); launch_minmax_f32(d_x, d_mask, nb, use_abs, center, reinterpret_cast<MinMaxF*>(small + kOffMinMax), ctx->stream); }
  MinMaxF mm;
  HIPCHK(copy_d2h(ctx, &mm, small + kOffMinMax, sizeof(mm)));
  HIPCHK(hipMemsetAsync(small + kOffMinMax, 0, sizeof(MinMaxF), ctx->stream));   // "nothing seen" again for the next user
  HIPCHK(CTX_SYNC());
  if (mm.min_inv == 0u) { *count = 0; *med = 0; return RSI_OK; }
  if (mm.nonfinite) return fail(ctx, RSI_ERR_UNSUPPORTED, "non-finite value in the transformed bins");
  auto unkey = [](uint32_t k) { uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &b, 4); return f; };
  const double ymin = unkey(~mm.min_inv), ymax = unkey(mm.max_bits);
  if ((ymax - ymin) < 0.01) return selection_mean(ctx, d_x, d_mask, nb, use_abs, center, med, count);
  const size_t np = (size_t)((ymax - ymin) / 0.01 + 2);
  if (np > (64u << 20)) return fail(ctx, RSI_ERR_UNSUPPORTED, "transformed bin range too wide for the 0.01 grid");
  HIPCHK(ctx->hist_f.ensure(np * 4));
  HIPCHK(hipMemsetAsync(ctx->hist_f.p, 0, np * 4, ctx->stream));
  { Timer t(ctx, "hist_f32"); launch_hist_f32(d_x, d_mask, nb, use_abs, center, ymin, ctx->hist_f.as<uint32_t>(), (uint32_t)np, ctx->stream); }
  std::vector<uint32_t> h(np);
  HIPCHK(copy_d2h(ctx, h.data(), ctx->hist_f.p, np * 4));
  HIPCHK(CTX_SYNC());
  uint64_t total = 0;
  for (uint32_t c : h) total += c;
  *count = total;
  *med = rsih::hist_median_grid(h.data(), np, total, ymin);
  return RSI_OK;
}

// The device-side form (kernels_bin.hip): a median is a chain of two launches (min/max + plan, histogram + walk), a
// (median, MAD) pair four, the MAD taking its centre from the median through device memory; the pair's results come back
// in the pinned mailbox with the last launch, which also clears what the scan pass behind it accumulates into.
// A range wider than the resident histogram falls back to grid_median().
constexpr uint32_t kGridCap = 1u << 20;
constexpr int32_t kFsListCap = 1 << 20;   // marked bins the level sums' compact list holds (more: the host loop)
struct ChainOut { GridMedian g[2]; uint32_t rawmin_inv; uint32_t pad[15]; };   // what a chain leaves in the mailbox
GridChain grid_chain(rsi_ctx* ctx) {
  uint8_t* small = ctx->small.as<uint8_t>();
  return GridChain{reinterpret_cast<MinMaxF*>(small + kOffMinMax), ctx->hist_f.as<uint32_t>(), kGridCap,
                   reinterpret_cast<unsigned int*>(small + kOffDone) + kDoneBinSlot};
}
// what the scan pass `pass` needs cleared: the first-L arrays, its work block, the run-boundary counter
void scan_fill_list(rsi_ctx* ctx, int pass, int64_t nb, FillList& fl) {
  uint8_t* small = ctx->small.as<uint8_t>();
  uint32_t* d_first_del = ctx->first_del.as<uint32_t>();
  const int64_t nbpad = (nb + 3) & ~int64_t(3);
  fill_add(fl, d_first_del, (size_t)(nbpad + nb) * 4, 0xffffffffu);
  fill_add(fl, small + kOffScanPass + (size_t)pass * kScanPassBytes, kScanPassBytes, 0u);
  fill_add(fl, small + kOffCounters, 32, 0u);
  if (pass == 0 && ctx->fs_ws.p) fill_add(fl, ctx->fs_ws.p, level_sums_head_bytes(kMaxL), 0u);   // filterstatus' level sums: marked count, flag, level counts
}
// Issues the (median, MAD) pair of the selection (mask == 0 where given).  planned: the first link's min/max + plan has been
// done by the kernel that produced x (launch_nb_scale_minmax).  with_median = false: only the MAD around `center`
// (planned then refers to the MAD's first link).  Returns the mailbox record the last launch fills.
int grid_pair_issue(rsi_ctx* ctx, const float* d_x, const int32_t* d_mask, int64_t nb, bool with_median, double center, bool planned,
                    int fill_pass, ChainOut** out) {
  uint8_t* small = ctx->small.as<uint8_t>();
  GridMedian* d_g = reinterpret_cast<GridMedian*>(small + kOffGrid);
  const GridChain gc = grid_chain(ctx);
  ChainOut* slot = static_cast<ChainOut*>(mb_alloc(ctx, sizeof(ChainOut)));
  if (!slot) return fail(ctx, RSI_ERR_INTERNAL, "out of pinned mailbox memory");
  memset(slot, 0, sizeof(*slot));
  GateShared gs(ctx);
  if (with_median) {
    if (!planned) { Timer t(ctx, "minmax_plan"); launch_minmax_plan(d_x, d_mask, nb, 0, 0.0, nullptr, gc, d_g, ctx->stream); }
    { Timer t(ctx, "hist_walk"); launch_hist_walk(d_x, d_mask, nb, 0, 0.0, nullptr, gc, d_g, nullptr, nullptr, ctx->stream); }
    planned = false;
  }
  const double* d_center = with_median ? &d_g->med : nullptr;
  if (!planned) { Timer t(ctx, "minmax_plan"); launch_minmax_plan(d_x, d_mask, nb, 1, center, d_center, gc, d_g + 1, ctx->stream); }
  GridExport ex{};
  ex.src[0] = d_g; ex.dst[0] = slot->g; ex.bytes[0] = 2 * sizeof(GridMedian);
  ex.src[1] = small + kOffRawMin; ex.dst[1] = &slot->rawmin_inv; ex.bytes[1] = 4;
  FillList fl{};
  if (fill_pass >= 0) scan_fill_list(ctx, fill_pass, nb, fl);
  { Timer t(ctx, "hist_walk"); launch_hist_walk(d_x, d_mask, nb, 1, center, d_center, gc, d_g + 1, &ex, &fl, ctx->stream); }
  *out = slot;
  return RSI_OK;
}
// what grid_median() would have returned for this record (same tests, same order, same messages); the selection's own
// description is needed for the degenerate case (its mean, on the host)
struct Selection { const float* d_x; const int32_t* d_mask; int64_t nb; int use_abs; double center; };
int grid_result(rsi_ctx* ctx, const GridMedian& g, const Selection& sel, double* med, uint64_t* count) {
  if (g.flags & kGridEmpty) { *count = 0; *med = 0; return RSI_OK; }
  if (g.flags & kGridNonFinite) return fail(ctx, RSI_ERR_UNSUPPORTED, "non-finite value in the transformed bins");
  if (g.flags & kGridDegenerate) return selection_mean(ctx, sel.d_x, sel.d_mask, sel.nb, sel.use_abs, sel.center, med, count);
  *med = g.med; *count = g.count;
  return RSI_OK;
}

// Per-L thresholds on the exact window sum equivalent to the reference's score tests
// (rsi.cpp:1204-1205, 1234-1235 with runmeantp's float mean, wufunctions.cpp:625-627).
void scan_thresholds(double tmedian, double tlamda, int Lmax, std::vector<double>& del, std::vector<double>& dup) {
  // kScanPad entries past Lmax that no sum can reach: the kernel walks L in unrolled groups
  del.assign((size_t)Lmax + 1 + kScanPad, -INFINITY);
  dup.assign((size_t)Lmax + 1 + kScanPad, INFINITY);
  del[0] = -1.0;
  for (int L = 1; L <= Lmax; ++L) {
    const double dL = (double)L, sq = sqrt(dL);
    auto score = [&](double sum) { const float meanf = (float)(sum / dL); return ((double)meanf - tmedian) * sq; };
    del[L] = rsih::last_true([&](double s) { return !(score(s) > -tlamda); });
    dup[L] = rsih::first_true([&](double s) { return !(score(s) < tlamda); });
  }
}

// Collect (pos << 1 | is_end) boundary entries from the device into sorted [start, end] pairs.
int fetch_pairs(rsi_ctx* ctx, const uint64_t* d_list, const uint32_t* d_count, uint32_t cap, std::vector<Region>& out,
                bool end_exclusive) {
  // the count and the first entries travel together (one round trip for all but the longest lists)
  constexpr uint32_t kEager = 1024;
  uint32_t cnt = 0;
  std::vector<uint64_t> raw(kEager);
  HIPCHK(copy_d2h(ctx, &cnt, d_count, 4));
  HIPCHK(copy_d2h(ctx, raw.data(), d_list, (size_t)std::min(cap, kEager) * 8));
  HIPCHK(CTX_SYNC());
  out.clear();
  if (cnt == 0) return RSI_OK;
  if (cnt > cap) return fail(ctx, RSI_ERR_UNSUPPORTED, "boundary list overflow");
  raw.resize(cnt);
  if (cnt > kEager) {
    HIPCHK(copy_d2h(ctx, raw.data() + kEager, d_list + kEager, (size_t)(cnt - kEager) * 8));
    HIPCHK(CTX_SYNC());
  }
  std::vector<int64_t> s, e;
  for (uint64_t v : raw) ((v & 1) ? e : s).push_back((int64_t)(v >> 1));
  if (s.size() != e.size()) return fail(ctx, RSI_ERR_INTERNAL, "unbalanced run boundaries");
  std::sort(s.begin(), s.end());
  std::sort(e.begin(), e.end());
  for (size_t i = 0; i < s.size(); ++i) out.push_back({(int)s[i], (int)(e[i] - (end_exclusive ? 1 : 0))});
  return RSI_OK;
}

// Marked runs in the reference's sense (last run not emitted, Q11) from what k_resolve_runs left in the mailbox
// ([count, 0][first kEagerBounds boundary entries]) and, for longer lists, in the device list.
constexpr uint32_t kEagerBounds = 1024;
int runs_from_export(rsi_ctx* ctx, const uint32_t* slot, const uint64_t* d_list, std::vector<Region>& runs) {
  const uint32_t cnt = slot[0];
  runs.clear();
  if (cnt == 0) return RSI_OK;
  if (cnt > kMaxRunEntries) return fail(ctx, RSI_ERR_UNSUPPORTED, "boundary list overflow");
  std::vector<uint64_t> raw(cnt);
  memcpy(raw.data(), slot + 2, (size_t)std::min(cnt, kEagerBounds) * 8);
  if (cnt > kEagerBounds) {
    HIPCHK(copy_d2h(ctx, raw.data() + kEagerBounds, d_list + kEagerBounds, (size_t)(cnt - kEagerBounds) * 8));
    HIPCHK(CTX_SYNC());
  }
  std::vector<int64_t> s, e;
  for (uint64_t v : raw) ((v & 1) ? e : s).push_back((int64_t)(v >> 1));
  if (s.size() != e.size()) return fail(ctx, RSI_ERR_INTERNAL, "unbalanced run boundaries");
  std::sort(s.begin(), s.end());
  std::sort(e.begin(), e.end());
  for (size_t i = 0; i < s.size(); ++i) runs.push_back({(int)s[i], (int)e[i]});
  runs.pop_back();
  return RSI_OK;
}

// The run list for the kernels: in the kernel arguments when short, else uploaded (start[k] | end[k]).
struct RunArgs { RunsInline inl; bool use_inl = false; int32_t* d_start = nullptr; int32_t* d_end = nullptr; };
int prepare_runs(rsi_ctx* ctx, const std::vector<Region>& runs, RunArgs& ra) {
  const size_t k = runs.size();
  if (k <= (size_t)kRunsInline) {
    memset(&ra.inl, 0, sizeof(ra.inl));
    for (size_t i = 0; i < k; ++i) { ra.inl.se[2 * i] = runs[i].start; ra.inl.se[2 * i + 1] = runs[i].end; }
    ra.use_inl = true;
    return RSI_OK;
  }
  HIPCHK(ctx->run_se.ensure(k * 8 + 64));
  std::vector<int32_t> se(2 * k);
  for (size_t i = 0; i < k; ++i) { se[i] = runs[i].start; se[k + i] = runs[i].end; }
  HIPCHK(copy_h2d(ctx, ctx->run_se.p, se.data(), k * 8));   // parked in the mailbox: se may go out of scope
  ra.use_inl = false;
  ra.d_start = ctx->run_se.as<int32_t>();
  ra.d_end = ctx->run_se.as<int32_t>() + k;
  return RSI_OK;
}

// Candidate stages on the device: uploads the host-prepared jobs, launches one workgroup per job,
// returns the statistics.  Jobs are grouped so that one launch's scratch stays below kCandScratchBytes.
constexpr size_t kCandScratchBytes = size_t(512) << 20;
class DeviceTester : public rsih::NeighbourTester {
 public:
  DeviceTester(rsi_ctx* c, DepthRef rdc, int64_t n, double RDmedian) : ctx(c), d_rdc(rdc), N(n), median(RDmedian) {}
  double kernel_wait_ms = 0;
  int launches = 0;

  // Jobs, neighbour lists and results of the candidate kernels live in the pinned mailbox (mapped host memory): the kernels
  // read a few dozen bytes per workgroup from it and write their records into it, so a call is its launches and one wait --
  // no upload, no download.  Lists too long for the mailbox take the copying path.
  bool sharpen(std::vector<Candidate>& L) override {
    if (L.empty()) return true;
    const int nj = (int)L.size();
    EdgeJob* slot = static_cast<EdgeJob*>(mb_alloc(ctx, (size_t)nj * sizeof(EdgeJob)));
    std::vector<EdgeJob> spill;
    EdgeJob* jobs = slot;
    if (!slot) { spill.resize((size_t)nj); jobs = spill.data(); }
    for (int i = 0; i < nj; ++i) jobs[i] = {L[(size_t)i].start, L[(size_t)i].end, L[(size_t)i].type, 0};
    {
      Phase ph(ctx, "cand.ensure");
      if (nj > ctx->sharpen_ws_jobs) {
        const int cap = std::max(256, 2 * nj);
        if (!ok(ctx->sharpen_ws.ensure(sharpen_workspace_bytes(cap)))) return false;
        if (!ok(hipMemsetAsync(ctx->sharpen_ws.p, 0, sharpen_workspace_zero_bytes(cap), ctx->stream))) return false;
        ctx->sharpen_ws_jobs = cap;
      }
      if (!slot && !ok(ctx->cand_jobs.ensure((size_t)nj * sizeof(EdgeJob)))) return false;
    }
    GateShared gs(ctx);
    Phase ph(ctx, "cand.sharpen");
    EdgeJob* d_jobs = slot ? slot : ctx->cand_jobs.as<EdgeJob>();
    if (!slot && !ok(copy_h2d(ctx, ctx->cand_jobs.p, jobs, (size_t)nj * sizeof(EdgeJob)))) return false;
    for (int pass = 0; pass < 2; ++pass) {   // rsi.cpp:1876-1877
      Timer t(ctx, "sharpen_edges");
      launch_sharpen_edges(d_rdc, N, d_jobs, nj, ctx->sharpen_ws.p, ctx->sharpen_ws_jobs, ctx->stream);
    }
    if (!slot && !ok(copy_d2h(ctx, jobs, ctx->cand_jobs.p, (size_t)nj * sizeof(EdgeJob)))) return false;
    if (!wait()) return false;
    gs.release();
    for (int i = 0; i < nj; ++i) { L[(size_t)i].start = jobs[i].start; L[(size_t)i].end = jobs[i].end; }
    return true;
  }

  bool test(const std::vector<rsih::TestPlan>& plans, std::vector<rsih::TestStats>& stats, std::vector<int>& left_reach,
            std::vector<char>& okv) override {
    const size_t n = plans.size();
    stats.assign(n, rsih::TestStats{});
    left_reach.assign(n, 0);
    okv.assign(n, 0);
    // One workgroup per test (k_candidate_test) or the four-launch form that spreads a test over 17 workgroups.  A lone
    // chromosome gets its tests back 40 % sooner from the second; in a pool the first is better (35.3 against 38.1 ms per
    // genome): a few long-lived workgroups cost the per-base kernels of the other chromosomes next to nothing, hundreds of
    // full ones take their CUs.  RSI_HOT_CAND_SPLIT=0 / 1 overrides.
    // several workgroups per test when the chip has room for them: a stand-alone context, or a pool run over a few
    // chromosomes only (a rank's share of a sharded genome); one workgroup per test when a dozen chromosomes share the chip
    const char* split_env = getenv("RSI_HOT_CAND_SPLIT");
    const bool split = split_env ? atoi(split_env) != 0 : true;
    size_t first = 0;
    while (first < n) {
      std::vector<CandJob> jobs;
      std::vector<int32_t> chains;   // (start, end) pairs
      size_t iwords = 0, lwords = 0, last = first;
      for (; last < n; ++last) {
        const rsih::TestPlan& T = plans[last];
        if (T.capacity <= 0 || T.end < T.start || T.start < 0 || T.end >= N) break;   // left to the host path (ok stays 0)
        auto up4 = [](size_t x) { return (x + 3) & ~(size_t)3; };   // the kernel wants every piece on a 16-byte boundary
        const size_t iw = up4((size_t)std::max(T.top + 1, 0)) + up4((size_t)T.capacity) + up4((size_t)std::min(T.capacity, T.budget)) +
                          (split ? up4((size_t)T.capacity) : 0);   // the split form's right walk has its own slots
        const size_t lw = up4((size_t)T.capacity + 1);
        if (!jobs.empty() && (iwords + iw) * 4 + (lwords + lw) * 8 > kCandScratchBytes) break;
        CandJob J{};
        J.start = T.start; J.end = T.end; J.kind = T.kind; J.margin = T.margin; J.capacity = T.capacity; J.top = T.top;
        J.nleft = (int)T.left_chain.size(); J.nright = (int)T.right_chain.size();
        J.left_off = (int)(chains.size() / 2);
        for (const auto& iv : T.left_chain) { chains.push_back(iv.first); chains.push_back(iv.second); }
        J.right_off = (int)(chains.size() / 2);
        for (const auto& iv : T.right_chain) { chains.push_back(iv.first); chains.push_back(iv.second); }
        J.budget = T.budget; J.cut = T.cut; J.right_cap = T.right_cap;
        J.iscratch_off = (int64_t)iwords; J.lscratch_off = (int64_t)lwords;
        iwords += iw; lwords += lw;
        jobs.push_back(J);
      }
      if (jobs.empty()) { ++first; continue; }   // plans[first] was declined
      chains.push_back(0); chains.push_back(0);  // never an empty upload
      std::vector<CandOut> outs(jobs.size());
      {
        Phase ph(ctx, "cand.ensure");
        if (!ok(ctx->cand_jobs.ensure(jobs.size() * sizeof(CandJob))) || !ok(ctx->cand_chains.ensure(chains.size() * 4)) ||
            !ok(ctx->cand_outs.ensure(outs.size() * sizeof(CandOut))) || !ok(ctx->cand_i32.ensure(iwords * 4)) ||
            !ok(ctx->cand_i64.ensure(lwords * 8)))
          return false;
        if (split) {   // per-job records and folded histograms of the split form: zero when (re)allocated, left zero by every launch
          // (re)allocated = the capacity changed: the allocator may hand the freed address out again, with whatever lies behind
          // the old extent -- comparing pointers missed exactly that, and a longer job list then started from stale counters
          const size_t m0 = ctx->cand_mid.cap, h0 = ctx->cand_hist.cap;
          if (!ok(ctx->cand_mid.ensure(jobs.size() * sizeof(CandMid))) || !ok(ctx->cand_hist.ensure(jobs.size() * (size_t)kCandHistBins * 4))) return false;
          if (ctx->cand_mid.cap != m0 && !ok(hipMemsetAsync(ctx->cand_mid.p, 0, ctx->cand_mid.cap, ctx->stream))) return false;
          if (ctx->cand_hist.cap != h0 && !ok(hipMemsetAsync(ctx->cand_hist.p, 0, ctx->cand_hist.cap, ctx->stream))) return false;
        }
      }
      GateShared gs(ctx);
      Phase ph(ctx, "cand.test");
      // jobs + chains + results in one mailbox slot (the kernels read / write mapped host memory); copies when it is too long
      const size_t jb = (jobs.size() * sizeof(CandJob) + 63) & ~size_t(63), cb = (chains.size() * 4 + 63) & ~size_t(63);
      const size_t ob = outs.size() * sizeof(CandOut);
      unsigned char* slot = jb + cb + ob <= kMailboxMaxCopy ? static_cast<unsigned char*>(mb_alloc(ctx, jb + cb + ob)) : nullptr;
      const CandJob* d_jobs = ctx->cand_jobs.as<CandJob>();
      const void* d_chains = ctx->cand_chains.p;
      CandOut* d_outs = ctx->cand_outs.as<CandOut>();
      if (slot) {
        memcpy(slot, jobs.data(), jobs.size() * sizeof(CandJob));
        memcpy(slot + jb, chains.data(), chains.size() * 4);
        d_jobs = reinterpret_cast<const CandJob*>(slot); d_chains = slot + jb; d_outs = reinterpret_cast<CandOut*>(slot + jb + cb);
      } else {
        if (!ok(copy_h2d(ctx, ctx->cand_jobs.p, jobs.data(), jobs.size() * sizeof(CandJob)))) return false;
        if (!ok(copy_h2d(ctx, ctx->cand_chains.p, chains.data(), chains.size() * 4))) return false;
      }
      {
        Timer t(ctx, split ? "candidate_test" : "candidate_test_one_wg");
        if (split)
          launch_candidate_test_split(d_rdc, N, d_jobs, (int)jobs.size(), d_chains, ctx->cand_i32.as<int32_t>(),
                                      ctx->cand_i64.as<long long>(), median, ctx->cand_mid.as<CandMid>(), ctx->cand_hist.as<uint32_t>(),
                                      d_outs, ctx->stream);
        else
          launch_candidate_test(d_rdc, N, d_jobs, (int)jobs.size(), d_chains, ctx->cand_i32.as<int32_t>(),
                                ctx->cand_i64.as<long long>(), median, d_outs, ctx->stream);
      }
      if (!slot && !ok(copy_d2h(ctx, outs.data(), ctx->cand_outs.p, outs.size() * sizeof(CandOut)))) return false;
      if (!wait()) return false;
      if (slot) memcpy(outs.data(), d_outs, ob);
      gs.release();
      ph.stop();
      static const bool cand_dbg = getenv("RSI_HOT_CAND_DBG") && atoi(getenv("RSI_HOT_CAND_DBG")) != 0;   // every test's plan and what the device found, on stderr
      for (size_t k = 0; k < jobs.size(); ++k) {
        const CandOut& O = outs[k];
        if (cand_dbg)
          fprintf(stderr, "[cand] [%d,%d] kind %d margin %d cap %d top %d chains %d/%d cut %d -> flags %d nref %d nbody %d nwin %d reach %d/%d ref q %.6f %.6f %.6f s1 %.6f s2 %.6f\n",
                  jobs[k].start, jobs[k].end, jobs[k].kind, jobs[k].margin, jobs[k].capacity, jobs[k].top, jobs[k].nleft, jobs[k].nright, jobs[k].cut,
                  O.flags, O.nref, O.nbody, O.nwin, O.left_reach, O.right_reach, O.ref_q[0], O.ref_q[1], O.ref_q[2], O.ref_s1, O.ref_s2);
        if (O.flags != 0) continue;
        rsih::TestStats& S = stats[first + k];
        const double nb = (double)O.nbody, nw = (double)O.nwin;
        S.cnv_lqt = O.body_q[0]; S.cnv_med = O.body_q[1]; S.cnv_uqt = O.body_q[2];
        { const double mu = O.body_s1 / nb; S.cnv_var = O.body_s2 / nb - mu * mu; }     // variancetp, wufunctions.cpp:766-809
        S.ref_lqt = O.ref_q[0]; S.ref_med = O.ref_q[1]; S.ref_uqt = O.ref_q[2];
        { const double mu = O.ref_s1 / nw; S.ref_var = O.ref_s2 / nw - mu * mu; }
        left_reach[first + k] = O.left_reach;
        okv[first + k] = 1;
      }
      first += jobs.size();
    }
    return true;
  }

  bool range_sums(const std::vector<std::pair<int, int>>& ranges, std::vector<int64_t>& sums) override {
    sums.assign(ranges.size(), 0);
    if (ranges.empty()) return true;
    std::vector<int32_t> flat;
    for (const auto& r : ranges) {
      if (r.first < 0 || r.second >= N || r.second < r.first) return false;
      flat.push_back(r.first); flat.push_back(r.second);
    }
    static_assert(sizeof(long long) == sizeof(int64_t), "int64");
    const size_t fb = (flat.size() * 4 + 63) & ~size_t(63);
    unsigned char* slot = static_cast<unsigned char*>(mb_alloc(ctx, fb + ranges.size() * 8));
    GateShared gs(ctx);
    if (slot) {   // ranges and sums in the mailbox
      memcpy(slot, flat.data(), flat.size() * 4);
      long long* d_sums = reinterpret_cast<long long*>(slot + fb);
      { Timer t(ctx, "range_sums"); launch_range_sums(d_rdc, slot, (int)ranges.size(), d_sums, ctx->stream); }
      if (!wait()) return false;
      memcpy(sums.data(), d_sums, ranges.size() * 8);
      return true;
    }
    if (!ok(ctx->cand_chains.ensure(flat.size() * 4)) || !ok(ctx->cand_outs.ensure(ranges.size() * 8))) return false;
    if (!ok(copy_h2d(ctx, ctx->cand_chains.p, flat.data(), flat.size() * 4))) return false;
    { Timer t(ctx, "range_sums"); launch_range_sums(d_rdc, ctx->cand_chains.p, (int)ranges.size(), ctx->cand_outs.as<long long>(), ctx->stream); }
    if (!ok(copy_d2h(ctx, sums.data(), ctx->cand_outs.p, ranges.size() * 8))) return false;
    return wait();
  }

 private:
  rsi_ctx* ctx;
  DepthRef d_rdc;
  int64_t N;
  double median;
  bool ok(hipError_t e) {
    if (e == hipSuccess) return true;
    ctx->err = std::string("candidate stage: ") + hipGetErrorString(e);
    set_global_error(ctx->err);
    failed = true;
    return false;
  }
  bool wait() {
    const double t0 = now_ms();
    const bool r = ok(ctx_sync(ctx));
    kernel_wait_ms += now_ms() - t0;
    ++launches;
    return r;
  }
 public:
  bool failed = false;
};

// One rsistatus pass on the device (rsi.cpp:1191-1259) -> d_status (and d_copy).  Three launches and no round trip: the scan
// (thresholds in the kernel arguments), the stop levels of the two sweeps, the status values with the run boundaries.  The
// first-L arrays, the work block and the boundary counter were cleared by the last launch of the quantile chain in front.
// What the host wants from the pass lands in the pinned mailbox: *work_slot = [ScanPassOut | per-L counts], *runs_slot =
// [count | first boundaries].
// A scan longer than kMaxL (a computed length, rsi.cpp:1286-1289, beyond 10 400 bins: a large -threshold, a very noisy chromosome)
// takes the same launches with its work block in a buffer of its own (the resident one is sized for kMaxL), cleared here, and
// -- where the pinned mailbox cannot hold the per-L counts -- brought back by a copy into `spill`.
int scan_pass(rsi_ctx* ctx, int pass, const float* d_T, const int32_t* d_medint, int64_t nb, double RDmedian, double tmedian,
              double tlamda, int Lmax, int32_t* d_status, int32_t* d_copy, const uint32_t** work_slot, const uint32_t** runs_slot,
              std::vector<uint32_t>* spill) {
  uint8_t* small = ctx->small.as<uint8_t>();
  std::vector<double> del, dup;
  scan_thresholds(tmedian, tlamda, Lmax, del, dup);
  const size_t nthr = (size_t)Lmax + 1 + kScanPad;
  ScanThr inl;
  const bool use_inl = nthr <= (size_t)kThrInline;
  double* d_del = nullptr;
  double* d_dup = nullptr;
  if (use_inl) {
    for (size_t k = 0; k < (size_t)kThrInline; ++k) { inl.del[k] = k < nthr ? del[k] : -INFINITY; inl.dup[k] = k < nthr ? dup[k] : INFINITY; }
  } else {
    HIPCHK(ctx->thr.ensure(nthr * 16));
    d_del = ctx->thr.as<double>();
    d_dup = d_del + nthr;
    del.insert(del.end(), dup.begin(), dup.end());
    HIPCHK(copy_h2d(ctx, d_del, del.data(), 2 * nthr * 8));
  }
  uint32_t* work = reinterpret_cast<uint32_t*>(small + kOffScanPass + (size_t)pass * kScanPassBytes);
  if (Lmax > kMaxL) {
    const size_t dev_bytes = (64 + 2 * (size_t)scan_level_stride(Lmax) * 4 + 255) & ~size_t(255);
    HIPCHK(ctx->scan_work_big.ensure(2 * dev_bytes));
    work = reinterpret_cast<uint32_t*>(ctx->scan_work_big.as<uint8_t>() + (size_t)pass * dev_bytes);
    HIPCHK(hipMemsetAsync(work, 0, dev_bytes, ctx->stream));
    if (pass == 0) ctx->phases.push_back({"scan.long (Lmax > 10400)", 1.0});
  }
  uint32_t* d_first_del = ctx->first_del.as<uint32_t>();
  uint32_t* d_first_dup = d_first_del + ((nb + 3) & ~int64_t(3));
  unsigned int* d_done = reinterpret_cast<unsigned int*>(small + kOffDone) + kDoneBinSlot;
  uint32_t* d_count = reinterpret_cast<uint32_t*>(small + kOffCounters) + 4;
  const size_t work_bytes = 64 + ((size_t)scan_level_stride(Lmax) + (size_t)Lmax + 1) * 4;
  if (scan_tile_workspace_bytes(Lmax)) HIPCHK(ctx->scan_ws.ensure(scan_tile_workspace_bytes(Lmax)));   // scans too long for an LDS tile
  uint32_t* rslot = static_cast<uint32_t*>(mb_alloc(ctx, 8 + (size_t)kEagerBounds * 8));
  uint32_t* wslot = static_cast<uint32_t*>(work_bytes <= kMailboxBytes / 8 ? mb_alloc(ctx, work_bytes) : nullptr);
  const bool spilled = !wslot && Lmax > kMaxL && spill;
  if (spilled) spill->assign((work_bytes + 3) / 4, 0u);
  if ((!wslot && !spilled) || !rslot) return fail(ctx, RSI_ERR_INTERNAL, "out of pinned mailbox memory");
  ScanParams sp;
  sp.nb = nb; sp.Lmax = Lmax; sp.pad = 0; sp.tmedian = tmedian;
  sp.lim_del = RDmedian * 0.75; sp.lim_dup = RDmedian * 1.25;
  { Timer t(ctx, "rsi_scan"); launch_rsi_scan(d_T, d_medint, sp, d_del, d_dup, use_inl ? &inl : nullptr, d_first_del, d_first_dup, work, ctx->scan_tiles.as<uint32_t>(), ctx->scan_ws.p, ctx->stream); }
  { Timer t(ctx, "level_stop"); launch_level_stop(d_first_del, d_first_dup, nb, Lmax, work, ctx->runs.p, d_done + 2, wslot, wslot ? work_bytes : 0, ctx->stream); }
  if (spilled) { HIPCHK(hipMemcpyAsync(spill->data(), work, work_bytes, hipMemcpyDeviceToHost, ctx->stream)); wslot = spill->data(); }
  { Timer t(ctx, "resolve_runs"); launch_resolve_runs(d_first_del, d_first_dup, work + 2, nb, d_status, d_copy, ctx->runs.as<uint64_t>(), d_count, kMaxRunEntries, d_done + 3, rslot, kEagerBounds, ctx->stream); }
  *work_slot = wslot; *runs_slot = rslot;
  return RSI_OK;
}

// rsicnvnbn (rsi.cpp:1262-1360) / rsicnvmed (rsi.cpp:1402-1501) around the device scan.  `first` = the record the quantile
// chain in front of the first pass leaves in the mailbox (issued by the caller, behind the transform).
int run_scan(rsi_ctx* ctx, const rsi_params& P, bool use_med, const float* d_T, int64_t nb, double RDmedian,
             double factor, int LmaxBase, ChainOut* first, double nb_med_raw, double nb_del_raw, double nb_dup_raw, ScanOut& out) {
  const int32_t* d_medint = ctx->binmed.as<int32_t>();
  double tmedian, tsigma, tlamda, target, dev, absmed;
  uint64_t cnt;
  int rc, cal_max;
  float t0 = 0, t2 = 0;
  {
    Phase ph(ctx, "scan.quantiles");
    HIPCHK(CTX_SYNC());
    const GridMedian* g = first->g;
    if ((!use_med && (g[0].flags & kGridTooWide)) || (g[1].flags & kGridTooWide)) {   // host-driven path, any range
      GateShared gs(ctx);
      if (!use_med) {
        if ((rc = grid_median(ctx, d_T, nullptr, nb, 0, 0.0, &tmedian, &cnt)) != RSI_OK) return rc;
      } else {
        tmedian = RDmedian;
      }
      if ((rc = grid_median(ctx, d_T, nullptr, nb, 1, tmedian, &absmed, &cnt)) != RSI_OK) return rc;
    } else {
      tmedian = RDmedian;
      if (!use_med && (rc = grid_result(ctx, g[0], Selection{d_T, nullptr, nb, 0, 0.0}, &tmedian, &cnt)) != RSI_OK) return rc;
      // the device chained the MAD to its own median; had that one been degenerate (its record then carries no median), the
      // deviations are taken again around the mean just computed
      if (!use_med && (g[0].flags & kGridDegenerate)) rc = grid_median(ctx, d_T, nullptr, nb, 1, tmedian, &absmed, &cnt);
      else rc = grid_result(ctx, g[1], Selection{d_T, nullptr, nb, 1, tmedian}, &absmed, &cnt);
      if (rc != RSI_OK) return rc;
    }
    if (!use_med) {   // the scaled reference levels (bins 0 and 2), as k_nb_scale_mm derived them from the raw minimum
      const uint32_t key = ~first->rawmin_inv;
      float tminf;
      { const uint32_t b = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key; memcpy(&tminf, &b, 4); }
      const double tmin = tminf;
      const double med_nbt = nb_med_raw - tmin;
      t0 = (float)((nb_del_raw - tmin) / med_nbt * RDmedian);
      t2 = (float)(med_nbt / med_nbt * RDmedian);
      (void)nb_dup_raw;
    }
  }
  tsigma = absmed / 0.6745;
  tlamda = factor * tsigma;
  if (!use_med) {
    target = (t2 - t0) * sqrt(2.5);                 // float difference, as RDtrans[2]-RDtrans[0]
    tlamda = std::max(tlamda, target);
    const double dnb = fabsf(t2 - t0) + 0.0001;
    const double q = tlamda * 2 / dnb;
    cal_max = (int)(q * q);
    dev = tsigma * 3.0;
  } else {
    target = tmedian * sqrt(2.0);
    tlamda = std::max(tlamda, target);
    if (P.threshold > 0) tlamda = tmedian * P.threshold;
    const double q = tlamda * 4 / (tmedian + 0.001);
    cal_max = (int)(q * q);
    dev = tmedian * 0.6;
  }
  int Lmax = LmaxBase;
  if (Lmax < cal_max) Lmax = cal_max;
  out.tmedian1 = tmedian; out.tsigma1 = tsigma; out.tlamda1 = tlamda; out.Lmax = Lmax;   // the reference's Lmax, as its log prints it
  // More lengths than bins: the reference's sweeps run L = 1, 2, ... and the PROGRAM exits at L = nb + 1 (runmean refuses a
  // span beyond the array, wufunctions.cpp:589-596) -- unless the 20 % rule (rsi.cpp:1226, 1256) has ended the sweep before,
  // which on a chromosome with so few bins per length it usually has.  So the scan runs up to nb lengths, and a sweep that
  // gets there without having stopped is what the reference exits on.
  const bool clipped = Lmax > nb;
  if (clipped) Lmax = (int)nb;
  // Up to kMaxL (the reference's own ceiling without a larger computed length is 10000, at -m 1) everything the scan needs is
  // resident; beyond, the scan takes its long form (scan_pass; kernels_bin.hip: device-memory tiles, 32-bit staged indices past
  // 32 000, per-L counts in device memory past 20 000) and filterstatus' level sums run on the host.  kHardMaxL bounds the
  // scratch (a tile of 4 M bins is 440 MB per workgroup); the reference itself would need years for such a sweep.
  if (Lmax > kHardMaxL) return fail(ctx, RSI_ERR_UNSUPPORTED, "scan length Lmax beyond 4194304 bins");
  const bool long_scan = Lmax > kMaxL;
  std::vector<uint32_t> spill1, spill2;
  auto sweeps_stopped = [&](const uint32_t* w) -> bool {   // both sweeps of a pass ended by the 20 % rule (w: the pass's work block)
    for (int k = 0; k < 2; ++k) {
      const uint32_t* cnt = w + 16 + (size_t)k * scan_level_stride(Lmax);
      uint64_t cum = 0;
      for (uint32_t L = 0; L <= w[2 + k] && L <= (uint32_t)Lmax; ++L) cum += cnt[L];
      if (!((double)(int)cum / (double)(int)nb > 0.2)) return false;
    }
    return true;
  };

  int32_t* d_st1 = ctx->status1.as<int32_t>();
  int32_t* d_st1f = ctx->status1f.as<int32_t>();
  int32_t* d_st2 = ctx->status2.as<int32_t>();
  const uint32_t* wslot = nullptr;
  const uint32_t* rslot = nullptr;
  { Phase ph(ctx, "scan.pass"); GateShared gs(ctx); if ((rc = scan_pass(ctx, 0, d_T, d_medint, nb, RDmedian, tmedian, tlamda, Lmax, d_st1, d_st1f, &wslot, &rslot, &spill1)) != RSI_OK) return rc; }

  Phase ph_filter(ctx, "scan.filterstatus");
  // ---- filterstatus (rsi.cpp:948-1047): the per-level sums are float accumulations in index order (App. A Q13) -- computed
  // on the device by an exact parallel form of the sequential loop (kernels_fs.hip), so neither the transformed bins nor
  // the status array travel to the host; the edge trimming runs on the device too, one thread per run ----
  const int nlev = 2 * Lmax + 1;
  uint32_t* fs_slot = long_scan ? nullptr : static_cast<uint32_t*>(mb_alloc(ctx, (size_t)nlev * 8));
  if (!fs_slot && !long_scan) return fail(ctx, RSI_ERR_INTERNAL, "out of pinned mailbox memory");
  if (!long_scan) {
    GateShared gs(ctx);
    Timer t(ctx, "level_sums");
    launch_level_sums(d_T, d_st1, nb, Lmax, ctx->fs_ws.p, kMaxL, kFsListCap, ctx->fs_out.as<float>(),
                      reinterpret_cast<unsigned int*>(ctx->small.as<uint8_t>() + kOffDone) + kDoneBinSlot + 4, fs_slot, ctx->stream);
  }
  { Phase phc(ctx, "fs.wait"); HIPCHK(CTX_SYNC()); }
  if (clipped && !sweeps_stopped(wslot)) return fail(ctx, RSI_ERR_TOO_SMALL, "fewer bins than the scan length, and a sweep reached them all (the reference exits in runmean)");
  out.escapes += wslot[0]; out.inexact = wslot[1]; out.tiles_listed += wslot[8];
  out.level_log[0].assign(wslot + 16, wslot + 16 + Lmax + 1);
  out.level_log[1].assign(wslot + 16 + scan_level_stride(Lmax), wslot + 16 + scan_level_stride(Lmax) + Lmax + 1);
  out.stop_levels[0] = wslot[2]; out.stop_levels[1] = wslot[3];
  {
    Phase phs(ctx, "fs.sums");
    // status values lie in [-Lmax, Lmax].  The level range the reference works on is [min status, max status]: taken from
    // the counts afterwards.
    std::vector<float> wsum((size_t)nlev, 0.0f);
    std::vector<int> wcnt((size_t)nlev, 0);
    if (fs_slot) {
      memcpy(wsum.data(), fs_slot, (size_t)nlev * 4);
      memcpy(wcnt.data(), fs_slot + nlev, (size_t)nlev * 4);
    }
    if (long_scan || wcnt[(size_t)Lmax] < 0) {   // a long scan, or the device declined (negative / non-finite values, too many marked bins): the loop itself
      Phase phc(ctx, "fs.copy");
      HIPCHK(ctx->h_T.ensure((size_t)nb * 4));
      HIPCHK(ctx->h_status.ensure((size_t)nb * 4));
      HIPCHK(hipMemcpyAsync(ctx->h_T.p, d_T, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->h_status.p, d_st1, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(CTX_SYNC());
      const int* st = ctx->h_status.as<int>();
      const float* tv = ctx->h_T.as<float>();
      std::fill(wsum.begin(), wsum.end(), 0.0f);
      std::fill(wcnt.begin(), wcnt.end(), 0);
      float s0 = 0.0f;
      int n0 = 0;
      for (int64_t i = 0; i < nb; ++i) {
        const int sv = st[i];
        if (sv == 0) { s0 += tv[i]; ++n0; }
        else { wsum[(size_t)(sv + Lmax)] += tv[i]; ++wcnt[(size_t)(sv + Lmax)]; }
      }
      wsum[(size_t)Lmax] = s0; wcnt[(size_t)Lmax] = n0;
    }
    int lo = 0, hi = 0;
    { int a = 0, b = 2 * Lmax; while (a < b && wcnt[a] == 0) ++a; while (b > a && wcnt[b] == 0) --b; lo = a - Lmax; hi = b - Lmax; }
    const int nl = hi - lo + 1;
    std::vector<float> lsum(wsum.begin() + (lo + Lmax), wsum.begin() + (hi + Lmax + 1));
    std::vector<int> lcnt(wcnt.begin() + (lo + Lmax), wcnt.begin() + (hi + Lmax + 1));
    for (int l = 0; l < nl; ++l) if (lcnt[l] != 0) lsum[l] /= (double)lcnt[l];
    phs.stop();
    Phase phr(ctx, "fs.runs");
    if (lo <= 0 && -lo < nl) {
      const float m0 = lsum[-lo];
      int leveldel = lo, leveladd = hi;
      for (int l = 0; l < nl; ++l) if (lsum[l] < m0 - dev) { leveldel = l + lo; break; }
      for (int l = nl - 1; l >= 0; --l) if (lsum[l] > m0 + dev) { leveladd = l + lo; break; }
      {   // the table the reference writes to its log: level, bins, mean; then the two chosen levels (rsi.cpp:991-997)
        char line[128];
        out.fs_lines.clear();
        for (int l = 0; l < nl; ++l) if (lcnt[l] != 0) { snprintf(line, sizeof(line), "%d\t%d\t%g", l + lo, lcnt[l], (double)lsum[l]); out.fs_lines.push_back(line); }
        snprintf(line, sizeof(line), "%d\t%g", leveldel, (double)lsum[leveldel - lo]); out.fs_lines.push_back(line);
        snprintf(line, sizeof(line), "%d\t%g", leveladd, (double)lsum[leveladd - lo]); out.fs_lines.push_back(line);
        if (leveldel > 0 || leveladd < 0 || leveldel > leveladd) out.fs_lines.push_back("warning level error, status not filtered");
      }
      if (!(leveldel > 0 || leveladd < 0 || leveldel > leveladd)) {
        std::vector<Region> runs;
        GateShared gs(ctx);
        if ((rc = runs_from_export(ctx, rslot, ctx->runs.as<uint64_t>(), runs)) != RSI_OK) return rc;
        if (!runs.empty()) {
          RunArgs ra;
          if ((rc = prepare_runs(ctx, runs, ra)) != RSI_OK) return rc;
          Timer t(ctx, "trim_runs");
          launch_trim_runs(d_T, d_st1f, ra.d_start, ra.d_end, ra.use_inl ? &ra.inl : nullptr, (int)runs.size(), (double)m0 - dev, (double)m0 + dev, ctx->stream);
        }
      }
    }
  }
  ph_filter.stop();
  // ---- second-pass parameters on the unmarked bins (rsi.cpp:1307-1319 / 1457-1469) ----
  Phase ph_q2(ctx, "scan.quantiles");
  double tmed2;
  uint64_t k = 0;
  ChainOut* second = nullptr;
  if ((rc = grid_pair_issue(ctx, d_T, d_st1f, nb, true, 0.0, false, 1, &second)) != RSI_OK) return rc;
  HIPCHK(CTX_SYNC());
  const GridMedian* g2 = second->g;
  {
    GateShared gs_q2(ctx);
    const bool wide2 = (g2[0].flags & kGridTooWide) != 0;
    if ((rc = wide2 ? grid_median(ctx, d_T, d_st1f, nb, 0, 0.0, &tmed2, &k) : grid_result(ctx, g2[0], Selection{d_T, d_st1f, nb, 0, 0.0}, &tmed2, &k)) != RSI_OK) return rc;
    if (k > (uint64_t)(nb / 2)) {
      tmedian = tmed2;
      if ((rc = (wide2 || (g2[1].flags & (kGridTooWide)) || (g2[0].flags & kGridDegenerate)) ? grid_median(ctx, d_T, d_st1f, nb, 1, tmedian, &absmed, &cnt)
                                                        : grid_result(ctx, g2[1], Selection{d_T, d_st1f, nb, 1, tmedian}, &absmed, &cnt)) != RSI_OK) return rc;
      tsigma = absmed / 0.6745;
      tlamda = factor * tsigma;
      tlamda = std::max(tlamda, target);
    }
  }
  out.tmedian2 = tmedian; out.tsigma2 = tsigma; out.tlamda2 = tlamda;
  ph_q2.stop();
  { Phase ph(ctx, "scan.pass"); GateShared gs(ctx); if ((rc = scan_pass(ctx, 1, d_T, d_medint, nb, RDmedian, tmedian, tlamda, Lmax, d_st2, nullptr, &wslot, &rslot, &spill2)) != RSI_OK) return rc; }
  Phase ph_seg(ctx, "scan.segments");
  GateShared gs_seg(ctx);

  // ---- get_rsi_segments (rsi.cpp:1060-1117) ----
  HIPCHK(CTX_SYNC());
  std::vector<Region> runs;
  if ((rc = runs_from_export(ctx, rslot, ctx->runs.as<uint64_t>(), runs)) != RSI_OK) return rc;
  if (clipped && !sweeps_stopped(wslot)) return fail(ctx, RSI_ERR_TOO_SMALL, "fewer bins than the scan length, and a sweep reached them all (the reference exits in runmean)");
  out.escapes += wslot[0]; out.inexact = wslot[1]; out.tiles_listed += wslot[8];
  out.level_log[2].assign(wslot + 16, wslot + 16 + Lmax + 1);
  out.level_log[3].assign(wslot + 16 + scan_level_stride(Lmax), wslot + 16 + scan_level_stride(Lmax) + Lmax + 1);
  out.stop_levels[2] = wslot[2]; out.stop_levels[3] = wslot[3];
  out.status2 = rsih::IntSpan();
  out.segs.clear();
  if (runs.empty()) return RSI_OK;
  std::vector<int64_t> poff(runs.size() + 1, 0);
  std::vector<SegItem> items;
  // About 256 (L, offset) pairs per thread of a workgroup in a pool that shares the chip (a long, thin kernel there costs the
  // other chromosomes next to nothing).  A chromosome that has the chip to itself (a stand-alone context, a pool run over a
  // few chromosomes) waits for exactly this kernel: as many items as still ride in the kernel arguments (kItemsInline), down
  // to 32 pairs per thread -- a 60 Mb chromosome's 2 M pairs then spread over ~130 workgroups instead of ~35.
  int64_t pairs_per_item = 1 << 16;
  if (!ctx->gate || ctx->gate->lonely()) {
    int64_t total = 0;
    for (const Region& r : runs) { const int64_t len = r.end - r.start + 1; total += len * (len + 1) / 2; }
    const int64_t room = std::max<int64_t>(8, (int64_t)kItemsInline - 2 * (int64_t)runs.size());   // every run's last item is a partial one
    pairs_per_item = std::min<int64_t>(pairs_per_item, std::max<int64_t>(1 << 13, total / room + 1));
  }
  const int64_t kPairsPerItem = pairs_per_item;
  for (size_t r = 0; r < runs.size(); ++r) {
    const int len = runs[r].end - runs[r].start + 1;
    poff[r + 1] = poff[r] + len + 1;
    int L = 1;
    while (L <= len) {   // chunks of lengths with about kPairsPerItem (L, offset) pairs
      int64_t pairs = 0; int Le = L;
      while (Le <= len && pairs < kPairsPerItem) { pairs += len - Le + 1; ++Le; }
      items.push_back({(int32_t)r, (int32_t)len, (int32_t)L, (int32_t)Le});
      L = Le;
    }
  }
  HIPCHK(ctx->scratch.ensure((size_t)poff.back() * 8 + (runs.size() + 1) * 8));
  double* d_scratch = ctx->scratch.as<double>();
  // short lists ride in the kernel arguments and the per-item results land in the pinned mailbox: two launches, no copy
  const bool inl = runs.size() <= (size_t)kRunsInline && items.size() <= (size_t)kItemsInline && poff.back() < (int64_t)1 << 31;
  std::vector<BestSeg> best(items.size());
  if (inl) {
    RunArgs ra;
    if ((rc = prepare_runs(ctx, runs, ra)) != RSI_OK) return rc;
    ItemsInline ii;
    memset(&ii, 0, sizeof(ii));
    for (size_t r = 0; r < runs.size(); ++r) { ra.inl.off[r] = (int32_t)poff[r]; ii.off[r] = (int32_t)poff[r]; }
    for (size_t i = 0; i < items.size(); ++i) ii.it[i] = items[i];
    BestSeg* slot = static_cast<BestSeg*>(mb_alloc(ctx, items.size() * sizeof(BestSeg)));
    if (!slot) return fail(ctx, RSI_ERR_INTERNAL, "out of pinned mailbox memory");
    // The host reads the status array inside the runs only (a segment's type, the nested levels of the block tests): the
    // prefix kernel writes those values, run after run, into the mailbox -- 2 % of the array instead of a copy of all of it.
    const size_t run_bins = (size_t)(poff.back() - (int64_t)runs.size());
    int32_t* sslot = run_bins * 4 <= kMailboxMaxCopy ? static_cast<int32_t*>(mb_alloc(ctx, std::max<size_t>(run_bins, 1) * 4)) : nullptr;
    if (!sslot) {
      HIPCHK(ctx->h_status2.ensure((size_t)nb * 4));
      HIPCHK(hipMemcpyAsync(ctx->h_status2.p, d_st2, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    { Timer t(ctx, "run_prefix"); launch_run_prefix(d_T, nullptr, nullptr, &ra.inl, (int)runs.size(), nullptr, d_scratch, d_st2, sslot, ctx->stream); }
    { Timer t(ctx, "best_subsegment"); launch_best_items(nullptr, &ii, (int)items.size(), nullptr, d_scratch, tmedian, slot, ctx->stream); }
    HIPCHK(CTX_SYNC());
    memcpy(best.data(), slot, items.size() * sizeof(BestSeg));
    if (sslot) {
      out.run_status.assign(sslot, sslot + run_bins);
      out.run_ranges.clear();
      for (size_t r = 0; r < runs.size(); ++r) out.run_ranges.push_back({runs[r].start, runs[r].end, poff[r] - (int64_t)r});
      out.status2 = rsih::IntSpan(out.run_status.data(), nb, &out.run_ranges);
    } else {
      out.status2 = rsih::IntSpan(ctx->h_status2.as<int>(), nb);
    }
  } else {
    RunArgs ra;
    const size_t kr = runs.size();
    HIPCHK(ctx->run_se.ensure(kr * 8 + 64));
    std::vector<int32_t> se(2 * kr);
    for (size_t i = 0; i < kr; ++i) { se[i] = runs[i].start; se[kr + i] = runs[i].end; }
    HIPCHK(copy_h2d(ctx, ctx->run_se.p, se.data(), kr * 8));
    int64_t* d_poff = reinterpret_cast<int64_t*>(d_scratch + poff.back());
    HIPCHK(ctx->items.ensure(items.size() * sizeof(SegItem)));
    HIPCHK(ctx->best.ensure(items.size() * sizeof(BestSeg)));
    HIPCHK(copy_h2d(ctx, d_poff, poff.data(), poff.size() * 8));
    HIPCHK(copy_h2d(ctx, ctx->items.p, items.data(), items.size() * sizeof(SegItem)));
    HIPCHK(ctx->h_status2.ensure((size_t)nb * 4));
    HIPCHK(hipMemcpyAsync(ctx->h_status2.p, d_st2, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
    out.status2 = rsih::IntSpan(ctx->h_status2.as<int>(), nb);   // complete at the sync below
    { Timer t(ctx, "run_prefix"); launch_run_prefix(d_T, ctx->run_se.as<int32_t>(), ctx->run_se.as<int32_t>() + kr, nullptr, (int)kr, d_poff, d_scratch, nullptr, nullptr, ctx->stream); }
    { Timer t(ctx, "best_subsegment"); launch_best_items(ctx->items.p, nullptr, (int)items.size(), d_poff, d_scratch, tmedian, ctx->best.as<BestSeg>(), ctx->stream); }
    HIPCHK(copy_d2h(ctx, best.data(), ctx->best.p, items.size() * sizeof(BestSeg)));
    HIPCHK(CTX_SYNC());
  }
  std::vector<BestSeg> per_run(runs.size(), BestSeg{-1.0, 0, 0});
  for (size_t i = 0; i < items.size(); ++i) {   // items of a run are in increasing L: strict > keeps the earliest
    BestSeg& b = per_run[(size_t)items[i].run];
    if (best[i].score > b.score) b = best[i];
  }
  for (size_t r = 0; r < runs.size(); ++r) {
    Candidate c;
    const int len = runs[r].end - runs[r].start + 1;
    double sc = per_run[r].score;
    if (sc > 0) { c.start = runs[r].start + per_run[r].start; c.end = c.start + per_run[r].len - 1; }
    else { c.start = runs[r].start; c.end = runs[r].start + len - 1; sc = 0; }
    const rsih::Quantiles q = rsih::grid_quantiles(out.status2.at(c.start), (size_t)(c.end - c.start + 1));
    if (q.med > 0) { c.type = rsih::kDup; c.score = sc; } else { c.type = rsih::kDel; c.score = -sc; }
    if (fabs(c.score) < tlamda * 0.5) continue;   // rsi.cpp:1343-1346
    out.segs.push_back(c);
  }
  return RSI_OK;
}

void to_call(const Candidate& c, rsi_call* o) {
  memset(o, 0, sizeof(*o));
  o->start = c.start; o->end = c.end; o->type = c.type; o->geno = c.geno; o->status = c.status; o->length = c.length;
  const double q1 = c.p1 < 1.0E-10 ? 99 : -10.0 * log(c.p1) / log(10.0);   // cnv_format1, rsi.cpp:583-585
  o->qscore = (int)q1;
  o->score = c.score; o->p1 = c.p1; o->cnvmed = c.cnvmed; o->cnvsd = c.cnvsd; o->cnviqr = c.cnviqr;
  o->refmed = c.refmed; o->refsd = c.refsd; o->refiqr = c.refiqr;
}

// What the per-base phase leaves for the bin-level stages.
struct PerBase {
  std::vector<Region> noncode;       // padded, merged N regions (reference coordinates)
  int64_t ncompact = 0, nb = 0;
  size_t res_vals = 0;               // values covered by the residue-class histogram
  std::vector<uint32_t> hres_all;    // BinAccum header + [value][MAD residue class] counts of the compacted depth
  double RDmedian = 0;
  bool host_stats = false;           // depths of 65 536 and more in the compacted array: the statistics came from the array itself
  double mads[31] = {};              // ... and so did the 31 subsamples' MADs (bin_level_stages)
};
constexpr size_t kResHead = 256;     // BinAccum sits in a header of the residue-class histogram: cleared and fetched with it

// ---- statistics of an int32 array on the host: the slow, exact path for depths the histograms do not cover --------------------
// The device's integer statistics are walks over directly indexed histograms of 65 536 values (kHistValues).  A chromosome
// whose median depth lies above that (or, without a cap, any value) used to be refused; now the array itself comes to the host
// -- 4 bytes per base over PCIe, a selection instead of a histogram walk: seconds for a large chromosome, and exact.
// value at which the cumulated count first reaches `rank` (partition_stat_tp's walk with dy = 1, wufunctions.cpp:398-420, as
// hist_quantiles_int restates it): the rank-th smallest, the minimum for rank 0 or when all values are equal
static double rank_value_i32(std::vector<int32_t>& v, uint64_t rank) {
  if (v.empty()) return 0.0;
  const auto mm = std::minmax_element(v.begin(), v.end());
  const int32_t lo = *mm.first, hi = *mm.second;
  if ((double)hi - (double)lo < 1.0 || rank == 0) return (double)lo;
  const size_t k = (size_t)std::min<uint64_t>(rank, v.size()) - 1;
  std::nth_element(v.begin(), v.begin() + k, v.end());
  return (double)v[k];
}
// the device array d_src[0 .. n) on the host
static int fetch_i32(rsi_ctx* ctx, const int32_t* d_src, int64_t n, std::vector<int32_t>& out) {
  out.resize((size_t)n);
  HIPCHK(hipMemcpyAsync(out.data(), d_src, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(CTX_SYNC());
  return RSI_OK;
}

// The GC-rescaled int32 array (what the reference leaves in RD after checkgccontent, gccontent.cpp:95): the run streams the
// byte copy and never writes it, so it is built when somebody asks -- rsi_hot_fetch("rd_gc"), or K4 on the configurations K4'
// does not cover -- from the last run's depth (still the caller's buffer), mask and GC table (still in the workspace).
int materialize_rd_gc(rsi_ctx* ctx) {
  if (ctx->rd_gc_valid) return RSI_OK;
  if (!ctx->have_gc || !ctx->last_depth) return fail(ctx, RSI_ERR_BAD_ARG, "no GC-adjusted run to build rd_gc from");
  uint8_t* small = ctx->small.as<uint8_t>();
  HIPCHK(ctx->rd_gc.ensure((size_t)(ctx->n + 4) * 4));
  { Timer t(ctx, "gc_materialize", true); launch_gc_materialize(ctx->last_depth, ctx->gcbits.as<uint64_t>(), ctx->n, reinterpret_cast<double*>(small + kOffTable), ctx->rd_gc.as<int32_t>(), reinterpret_cast<unsigned int*>(small + kOffDone) + 4 * kDoneStride, ctx->stream); }
  ctx->rd_gc_valid = true;
  return RSI_OK;
}

// The capped, compacted depth as int32 (the reference's RD after concatenate_data): K4' leaves it as bytes, which is what the
// candidate kernels read; the int32 form is built when somebody asks -- rsi_hot_fetch("rd_concat"), the host's page fetches.
int materialize_rdc(rsi_ctx* ctx) {
  if (!ctx->rdc_is_bytes || ctx->rdc_valid) return RSI_OK;
  HIPCHK(ctx->rdc.ensure((size_t)(ctx->ncompact + 4) * 4));
  { Timer t(ctx, "rdc_widen", true); launch_widen_u8(ctx->rdc8.as<uint8_t>(), ctx->ncompact, ctx->rdc.as<int32_t>(), ctx->stream); }
  ctx->rdc_valid = true;
  return RSI_OK;
}

// A1-A9: GC mask and N runs, GC table and rescale, cap, compaction, bins, chromosome statistics (K1-K4).  The kernels are
// HBM-bound: workers of a pool take turns through this phase (GpuGate, held until the function returns).
int per_base_phase(rsi_ctx* ctx, const rsi_params& P, const int32_t* d_depth, const uint8_t* d_fasta, int64_t n, rsi_result* res,
                   PerBase& pb) {
  std::vector<Region>& noncode = pb.noncode;
  // The per-base kernels are HBM-bound: workers of a pool take turns through this phase (GpuGate).
  struct StreamTurn {
    GpuGate* g = nullptr;
    ~StreamTurn() { if (g) g->unlock(); }
    void release() { if (g) { g->unlock(); g = nullptr; } }
  } hbm_turn;
  if (ctx->gate) {
    Phase ph_wait(ctx, "wait.hbm_turn");
    ctx->gate->lock(ctx->gate_shared);
    hbm_turn.g = ctx->gate;
  }
  Phase ph_a1(ctx, "a1.classify+nruns");
  ctx->n = n; ctx->ncompact = 0; ctx->nb = 0; ctx->have_gc = ctx->have_nb = ctx->have_med = false;
  ctx->rd_gc_valid = false; ctx->last_depth = d_depth; ctx->rdc_is_bytes = false; ctx->rdc_valid = false;
  rsi_chrom_stats& S = res->stats;
  memset(&S, 0, sizeof(S));
  S.n = n;
  res->params = P;
  hipStream_t st = ctx->stream;
  const int64_t nwords = n / 64 + 1;

  Phase ph_a1a(ctx, "a1a.ensure");
  HIPCHK(ctx->small.ensure(kSmallBytes));
  HIPCHK(ctx->gcbits.ensure((size_t)nwords * 8));
  HIPCHK(ctx->nbits.ensure((size_t)nwords * 8));
  HIPCHK(ctx->hist_val.ensure((size_t)kHistValues * 4));
  HIPCHK(ctx->gsum.ensure(fold_scratch_bytes()));
  static_assert(sizeof(BinAccum) <= kResHead, "BinAccum outgrew its header");
  HIPCHK(ctx->hist_res.ensure(kResHead + (size_t)kHistValues * kResClasses * 4));
  uint8_t* small = ctx->small.as<uint8_t>();
  unsigned int* d_done = reinterpret_cast<unsigned int*>(small + kOffDone);   // arrival counters: K2, K3, K4
  ph_a1a.stop();
  Phase ph_a1b(ctx, "a1b.launch K1-K3");
  const bool want_cap = P.cap > 1;
  // Two per-base passes behind K1 instead of three: K2j counts (GC count, depth byte) pairs, its last workgroup derives the GC
  // table, the rescaled-value histogram and the cap median from them, K4j rescales on its way to the bins.  RSI_HOT_JOINT=0:
  // the three-pass chain (K2, K3', K4'), which is also where a chromosome goes whose joint counters wrapped.
  const char* joint_env = getenv("RSI_HOT_JOINT");
  const bool joint = P.gcadjust && want_cap && !(joint_env && atoi(joint_env) == 0);
  // -NOGC with a cap: the histogram pass leaves the depth as bytes and (round 5) the cap on the device, so that K4s / K4m can be queued behind it
  const bool nogc_bytes = !P.gcadjust && want_cap && !(getenv("RSI_HOT_NOGC_BYTES") && atoi(getenv("RSI_HOT_NOGC_BYTES")) == 0);
  // one buffer: the folded pair counters (cleared by K1) | the workgroups' escape lists | the levels' fixed-point ratios for K4j
  const size_t joint_list_off = (gc_joint_totals_bytes() + 255) & ~size_t(255), joint_lut_off = joint_list_off + ((gc_joint_esc_list_bytes() + 255) & ~size_t(255));
  if (joint) HIPCHK(ctx->joint_tot.ensure(joint_lut_off + (size_t)kGcLevels * 4 + 64));

  // ---- A1-A4 are issued back to back: GC mask and N runs (K1, K1b), GC table and rescale (K2, K3), the cap
  // median walk.  K1 clears the accumulators of everything behind it, K2's last workgroup builds the GC table, K3's
  // last workgroup walks the value histogram to the median and writes the header (N-run list, GC accumulators for the
  // checks and the log, median) into the pinned mailbox: four launches, no memset, no copy, ONE round trip. ----
  uint32_t* d_ncount = reinterpret_cast<uint32_t*>(small + kOffCounters) + 5;
  {
    FillList fl{};
    fill_add(fl, small, kHeaderBytes, 0u);
    if (P.gcadjust || want_cap) fill_add(fl, ctx->hist_val.p, (size_t)kHistValues * 4, 0u);
    fill_add(fl, ctx->hist_res.p, kResHead + (size_t)256 * kResClasses * 4, 0u);   // BinAccum of K4 and the rows K4' adds its groups' sums to (its range is at most 256 values)
    if (joint) fill_add(fl, ctx->joint_tot.p, gc_joint_totals_bytes(), 0u);       // K2j's folded joint histogram
    Timer t(ctx, "fasta_classify", true);
    launch_fasta_classify(d_fasta, n, ctx->gcbits.as<uint64_t>(), ctx->nbits.as<uint64_t>(), nwords, fl, st);
  }
  uint64_t* d_ntrans = reinterpret_cast<uint64_t*>(small + kOffNtrans);
  PhaseParams* d_pp = reinterpret_cast<PhaseParams*>(small + kOffPhase);
  int64_t* d_cbreak = reinterpret_cast<int64_t*>(small + kOffBreaks);
  int64_t* d_cum = d_cbreak + 4100;
  // K1b: the N runs' boundaries and the removed regions -- inside K2j's launch where there is one (RSI_HOT_K1B_INSIDE=1; measured: 0.7 % on the pooled step, 13 us of a lone
  // chromosome -- and 10 us MORE for K2j itself, the kernel the roofline is quoted on: off by default)
  const bool k1b_inside = joint && getenv("RSI_HOT_K1B_INSIDE") && atoi(getenv("RSI_HOT_K1B_INSIDE")) == 1;
  const NRuns k1b_args{ctx->nbits.as<uint64_t>(), d_ntrans, d_ncount, (uint32_t)kMaxTransitions, std::max(50, P.m / 4), d_cbreak, d_cum};
  if (!k1b_inside) { Timer t(ctx, "n_transitions"); launch_n_transitions(ctx->nbits.as<uint64_t>(), nwords, d_ntrans, d_ncount, kMaxTransitions, n, std::max(50, P.m / 4), (joint || nogc_bytes) ? d_pp : nullptr, d_cbreak, d_cum, d_done + 4 * kDoneStride + 8, st); }
  constexpr uint32_t kEagerRuns = 1024;
  uint32_t n_trans = 0;
  std::vector<uint64_t> trans_raw(kEagerRuns);
  // the header (accumulators, counters, median) and the first entries of the N-run list: one block, written by the last kernel of the chain
  const size_t head_bytes = kHeaderBytes + (size_t)kEagerRuns * 8;
  unsigned char* head = static_cast<unsigned char*>(mb_alloc(ctx, head_bytes));
  if (!head) return fail(ctx, RSI_ERR_INTERNAL, "out of pinned mailbox memory");

  const int32_t* d_src = d_depth;
  ValueHistAux* d_aux = reinterpret_cast<ValueHistAux*>(small + kOffValAux);
  GcAccum* d_acc = reinterpret_cast<GcAccum*>(small + kOffGcAcc);
  double* d_table = reinterpret_cast<double*>(small + kOffTable);
  ValueMedian* d_vm = reinterpret_cast<ValueMedian*>(small + kOffValMedian);
  GcAccum acc;
  ValueHistAux aux;
  ValueMedian vm;
  JointInfo jinfo;
  memset(&acc, 0, sizeof(acc)); memset(&aux, 0, sizeof(aux)); memset(&vm, 0, sizeof(vm)); memset(&jinfo, 0, sizeof(jinfo));
  // packed = 1 first; depths of 2^21 and more make the packed accumulators overflow (flag bit 1): everything from K2 on is
  // then issued once more with the two-atomic form
  auto issue_joint = [&]() -> int {
    HIPCHK(ctx->depth8.ensure((size_t)n + 2048));
    Timer t(ctx, "gc_joint_hist", true);
    launch_gc_joint_hist(d_depth, ctx->gcbits.as<uint64_t>(), n, d_acc, d_table, ctx->slabs.p, ctx->joint_tot.p, d_done, ctx->depth8.as<uint8_t>(),
                         ctx->hist_val.as<uint32_t>(), d_aux, d_vm, small, head, head_bytes, ctx->joint_tot.as<uint8_t>() + joint_list_off,
                         reinterpret_cast<unsigned int*>(ctx->joint_tot.as<uint8_t>() + joint_lut_off), reinterpret_cast<JointInfo*>(small + kOffJointInfo),
                         d_pp, (double)P.cap, st, k1b_inside ? &k1b_args : nullptr);
    return RSI_OK;
  };
  // K4j right behind K2j, no host in between (one wait per per-base phase instead of two): regions, length and cap reach it
  // through device memory (K1b's and K2j's last workgroups); what the launch itself must know -- the value range of its LDS
  // histogram, the median phase's packing -- comes from the cap of this context's previous chromosome under the same flags.
  // The kernel declines when the real cap does not fit that configuration; the host checks everything again below and
  // launches the ordinary way whenever anything differs.  RSI_HOT_SPEC=0 switches it off.
  const char* spec_env = getenv("RSI_HOT_SPEC");
  const bool k4j_fix_off = getenv("RSI_HOT_K4J_FIX") && atoi(getenv("RSI_HOT_K4J_FIX")) == 0;   // the queued K4j always takes K2j's ratios
  // K4 as three launches (kernels_k4s.hip: streaming half at eight waves per SIMD, the odd chunks exactly, the bin medians without
  // LDS) wherever K2j's verified ratios exist; RSI_HOT_K4SPLIT=0: K4j, the one-kernel form
  const bool k4_split = !(getenv("RSI_HOT_K4SPLIT") && atoi(getenv("RSI_HOT_K4SPLIT")) == 0);
  const bool spec_ok = !(spec_env && atoi(spec_env) == 0) && ctx->spec_capval >= 1 && ctx->spec_m == P.m && ctx->spec_cap == (double)P.cap &&
                       ctx->spec_gc == (P.gcadjust != 0) && cap_compact8_applies(P.m, ctx->spec_capval);
  const bool spec_nogc = nogc_bytes && spec_ok && k4_split && rescale_compact_split_applies(P.m, ctx->spec_capval, n, 0);   // (the split form only)
  const bool spec = (joint && spec_ok && !k4j_fix_off) || spec_nogc;
  constexpr uint32_t kSpecMagic = 0x5bec5bec;
  uint32_t* spec_slot = nullptr;
  size_t spec_bytes = 0;
  auto issue_k4j_spec = [&]() -> int {
    const int32_t guess = ctx->spec_capval;
    int vr = 64;
    while (vr < 256 && vr <= guess) vr <<= 1;
    spec_bytes = kResHead + (size_t)vr * kResClasses * 4;
    spec_slot = static_cast<uint32_t*>(mb_alloc(ctx, spec_bytes));
    if (!spec_slot) return RSI_OK;   // no room in the mailbox: the ordinary way
    spec_slot[3] = kSpecMagic;       // BinAccum::pad: the kernel's export overwrites it with zero; still there = the kernel declined
    HIPCHK(ctx->slabs.ensure(std::max(gc_joint_slab_bytes(n), std::max(cap_compact8_slab_bytes(P.m, guess, n), rescale_compact_split_slab_bytes(guess, n)))));
    HIPCHK(ctx->rdc8.ensure(rescale_compact_split_rdc_bytes(n)));
    HIPCHK(ctx->binmed.ensure((size_t)(n / P.m + 1) * 4));
    HIPCHK(ctx->binsum.ensure((size_t)(n / P.m + 1) * 8));
    K4Regions none;
    memset(&none, 0, sizeof(none));
    Timer t(ctx, "cap_compact_bin", true);
    if (spec_nogc) {   // -NOGC: the bytes are the values (no ratios, no escapes: the histogram pass saturates them at 254, above any cap of this path)
      launch_rescale_compact_stream(ctx->rescaled8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, nullptr, d_cbreak, d_cum, none, 0, n, guess, P.m,
                                    ctx->rdc8.as<uint8_t>(), reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->hist_res.p) + kResHead), ctx->slabs.p,
                                    d_done + 2 * kDoneStride, ctx->hist_res.p, spec_slot, spec_bytes, nullptr, nullptr, d_pp, st);
      t.~Timer();
      new (&t) Timer(ctx, "bin_median", true);
      launch_bin_median8(ctx->rdc8.as<uint8_t>(), n, guess, P.m, ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), d_pp, st);
      return RSI_OK;
    }
    if (k4_split && rescale_compact_split_applies(P.m, guess, n, 0)) {
      launch_rescale_compact_stream(ctx->depth8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, d_cbreak, d_cum, none, 0, n, guess, P.m,
                                    ctx->rdc8.as<uint8_t>(), reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->hist_res.p) + kResHead), ctx->slabs.p,
                                    d_done + 2 * kDoneStride, ctx->hist_res.p, spec_slot, spec_bytes,
                                    reinterpret_cast<const unsigned int*>(ctx->joint_tot.as<uint8_t>() + joint_lut_off), &d_acc->escapes, d_pp, st);
      t.~Timer();   // (closes the streaming half's bracket: the medians have their own)
      new (&t) Timer(ctx, "bin_median", true);
      launch_bin_median8(ctx->rdc8.as<uint8_t>(), n, guess, P.m, ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), d_pp, st);
      return RSI_OK;
    }
    launch_rescale_compact_bin8(ctx->depth8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, d_cbreak, d_cum, none, 0, n, guess, P.m,
                                ctx->rdc8.as<uint8_t>(), ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(),
                                reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->hist_res.p) + kResHead), ctx->slabs.p, ctx->gsum.p, d_done + 2 * kDoneStride,
                                ctx->hist_res.p, spec_slot, spec_bytes, reinterpret_cast<const unsigned int*>(ctx->joint_tot.as<uint8_t>() + joint_lut_off), d_pp, st);
    return RSI_OK;
  };
  auto issue_gc_chain = [&](int packed) -> int {
    if (P.gcadjust) {
      // K2 leaves a byte copy of the depth; K3' streams that copy (1 byte per base instead of 4) and writes nothing per base
      HIPCHK(ctx->depth8.ensure((size_t)n + 2048));
      HIPCHK(ctx->rescaled8.ensure((size_t)n + 2048));
      { Timer t(ctx, packed ? "gc_hist" : "gc_hist_wide", true); launch_gc_hist(d_depth, ctx->gcbits.as<uint64_t>(), n, d_acc, d_table, packed, ctx->slabs.p, ctx->gsum.p, d_done, ctx->depth8.as<uint8_t>(), st); }
      { Timer t(ctx, "value_hist8", true); launch_value_hist8(ctx->depth8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, ctx->hist_val.as<uint32_t>(), d_aux, ctx->slabs.p, ctx->gsum.p, d_done + kDoneStride, d_vm, small, head, head_bytes, ctx->rescaled8.as<uint8_t>(), &d_acc->escapes, st); }
    } else if (want_cap) {
      // -NOGC: the histogram pass for the cap median also leaves the depth as bytes (saturated at 254: K4' only needs
      // min(value, cap) and is taken when the cap is below that), so that the compaction moves 1 + 1 bytes per base
      // instead of 4 + 4 (loaddata.cpp:229-240, 48-85 without gccontent.cpp in front)
      HIPCHK(ctx->slabs.ensure(gc_rescale_slab_bytes(n)));
      HIPCHK(ctx->rescaled8.ensure((size_t)n + 2048));
      { Timer t(ctx, "value_hist", true); launch_gc_rescale(d_depth, ctx->gcbits.as<uint64_t>(), n, nullptr, 0, nullptr, ctx->hist_val.as<uint32_t>(), d_aux, ctx->slabs.p, ctx->gsum.p, d_done + kDoneStride, d_vm, small, head, head_bytes, st, ctx->rescaled8.as<uint8_t>(), nogc_bytes ? d_pp : nullptr, (double)P.cap); }
    } else {
      HIPCHK(copy_d2h(ctx, head, small, head_bytes));   // no kernel behind K1b to hand the header over: a plain copy
    }
    return RSI_OK;
  };
  auto unpack_head = [&]() {
    memcpy(&acc, head + kOffGcAcc, sizeof(acc));
    memcpy(&jinfo, head + kOffJointInfo, sizeof(jinfo));
    memcpy(&aux, head + kOffValAux, sizeof(aux));
    memcpy(&vm, head + kOffValMedian, sizeof(vm));
    memcpy(&n_trans, head + kOffCounters + 5 * 4, 4);
    memcpy(trans_raw.data(), head + kOffNtrans, (size_t)kEagerRuns * 8);
  };
  // the slab buffer serves K2 and K3 one after the other: size it for both before anything is in flight
  if (P.gcadjust) HIPCHK(ctx->slabs.ensure(std::max(std::max(gc_hist_slab_bytes(n), joint ? gc_joint_slab_bytes(n) : 0), std::max(gc_rescale_slab_bytes(n), value_hist8_slab_bytes(n)))));
  if (spec) HIPCHK(ctx->slabs.ensure(std::max(gc_joint_slab_bytes(n), std::max(cap_compact8_slab_bytes(P.m, ctx->spec_capval, n), rescale_compact_split_slab_bytes(ctx->spec_capval, n)))));   // before anything is in flight
  int rc = joint ? issue_joint() : issue_gc_chain(1);
  if (rc != RSI_OK) return rc;
  if (spec && (rc = issue_k4j_spec()) != RSI_OK) return rc;
  ph_a1b.stop();
  { Phase ph_a1c(ctx, "a1c.wait K1-K3"); HIPCHK(CTX_SYNC()); }
  unpack_head();
  bool joint_ok = joint;
  if (joint && (acc.negatives & 4u)) {
    // a workgroup's 16-bit pair counters wrapped (a sequence without GC variation under a constant depth): the three-pass
    // chain from clean accumulators
    Phase ph_w(ctx, "a2-3.joint wrapped: three-pass chain");
    joint_ok = false;
    FillList fl{};
    fill_add(fl, small + kOffGcAcc, 4096, 0u);
    fill_add(fl, ctx->hist_val.p, (size_t)kHistValues * 4, 0u);
    fill_add(fl, d_aux, 16, 0u);
    fill_add(fl, d_vm, 32, 0u);
    launch_fill(fl, st);
    if ((rc = issue_gc_chain(1)) != RSI_OK) return rc;
    HIPCHK(CTX_SYNC());
    unpack_head();
  }
  if (joint_ok && !(acc.negatives & 1u) && jinfo.esc_pending && (uint64_t)acc.escapes <= (uint64_t)byte_escape_limit(n)) {
    // depths of 255 and more are not in K2j's pair counters, and there were too many for its last workgroup to add them from the
    // workgroups' lists: their rescaled values enter the histogram now, then the median walk
    Phase ph_e(ctx, "a2-3.escapes");
    { Timer t(ctx, "escape_hist", true); launch_escape_hist(ctx->depth8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, ctx->hist_val.as<uint32_t>(), d_aux, d_done + kDoneStride, d_vm, small, head, head_bytes, st); }
    HIPCHK(CTX_SYNC());
    unpack_head();
  }
  if (P.gcadjust && !joint_ok && (acc.negatives & 2u)) {
    Phase ph_w(ctx, "a2-3.gc wide redo");
    FillList fl{};
    fill_add(fl, ctx->hist_val.p, (size_t)kHistValues * 4, 0u);
    fill_add(fl, d_aux, 16, 0u);
    static_assert(sizeof(ValueHistAux) == 16, "ValueHistAux is cleared as one 16-byte unit");
    launch_fill(fl, st);
    if ((rc = issue_gc_chain(0)) != RSI_OK) return rc;
    HIPCHK(CTX_SYNC());
    unpack_head();
  }

  // Deep coverage: more than an eighth of the bases did not fit K2's byte copy, and K3' only handed the header over.  The
  // rescale, the value histogram and the cap median come from the int32 kernel instead (its LDS histogram follows the mean
  // depth, kernels.h: hist_window_base), which also leaves the rescaled int32 array for K4.
  bool deep = false;
  if (P.gcadjust && (uint64_t)acc.escapes > (uint64_t)byte_escape_limit(n)) {
    Phase ph_d(ctx, "a2-3.deep coverage");
    deep = true;
    if (joint_ok) {   // K2j's last workgroup has counted the byte-sized depths already: the int32 kernel starts from a clean histogram
      FillList fl{};
      fill_add(fl, ctx->hist_val.p, (size_t)kHistValues * 4, 0u);
      fill_add(fl, d_aux, 16, 0u);
      launch_fill(fl, st);
    }
    HIPCHK(ctx->rd_gc.ensure((size_t)(n + 4) * 4));
    HIPCHK(ctx->slabs.ensure(gc_rescale_slab_bytes(n)));
    { Timer t(ctx, "gc_rescale", true); launch_gc_rescale(d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, 1, ctx->rd_gc.as<int32_t>(), ctx->hist_val.as<uint32_t>(), d_aux, ctx->slabs.p, ctx->gsum.p, d_done + kDoneStride, d_vm, small, head, head_bytes, st); }
    HIPCHK(CTX_SYNC());
    unpack_head();
    ctx->rd_gc_valid = true;
  }

  // ---- N runs -> padded, merged regions (get_noseq_regions, loaddata.cpp:243-273) ----
  std::vector<Region> nruns;
  {
    if (n_trans > kMaxTransitions) return fail(ctx, RSI_ERR_UNSUPPORTED, "boundary list overflow");
    trans_raw.resize(n_trans > kEagerRuns ? n_trans : std::max<uint32_t>(n_trans, 0));
    if (n_trans > kEagerRuns) {
      HIPCHK(copy_d2h(ctx, trans_raw.data() + kEagerRuns, d_ntrans + kEagerRuns, (size_t)(n_trans - kEagerRuns) * 8));
      HIPCHK(CTX_SYNC());
    }
    std::vector<int64_t> rs, re;
    for (uint32_t i = 0; i < n_trans; ++i) { const uint64_t v = trans_raw[i]; ((v & 1) ? re : rs).push_back((int64_t)(v >> 1)); }
    if (rs.size() != re.size()) return fail(ctx, RSI_ERR_INTERNAL, "unbalanced run boundaries");
    std::sort(rs.begin(), rs.end());
    std::sort(re.begin(), re.end());
    for (size_t i = 0; i < rs.size(); ++i) nruns.push_back({(int)rs[i], (int)(re[i] - 1)});
  }
  {
    const int dx = std::max(50, P.m / 4);
    for (const Region& r : nruns) {
      Region g{std::max(0, r.start - dx), (int)std::min<int64_t>(n - 1, (int64_t)r.end + dx)};
      if (!noncode.empty() && g.start <= noncode.back().end + 1) noncode.back().end = std::max(noncode.back().end, g.end);
      else noncode.push_back(g);
    }
  }
  if ((int)noncode.size() > kMaxRegions) return fail(ctx, RSI_ERR_UNSUPPORTED, "more than 4096 N regions");
  S.n_noncode = (int)noncode.size();
  res->noncode.clear();
  for (const Region& r : noncode) { res->noncode.push_back(r.start); res->noncode.push_back(r.end); }
  ph_a1.stop();

  Phase ph_cap(ctx, "a4.checks+cap");
  if (P.gcadjust) {
    if (acc.negatives & 1u) return fail(ctx, RSI_ERR_UNSUPPORTED, "negative depth values");
    double rdmean = (double)acc.possum;                       // gccontent.cpp:109-112 (the device's k_gc_table computes the same)
    if (acc.poscnt > 0) rdmean /= (double)acc.poscnt;
    S.gc_rdmean = rdmean;
    S.byte_escapes = acc.escapes;
    ctx->have_gc = true;   // rsi_hot_fetch builds the rescaled int32 array on demand (materialize_rd_gc)
  }
  // ---- A4: cap from the median of the uncompacted array (loaddata.cpp:229-240, Q15) ----
  int32_t capval = 0x7fffffff;
  if (want_cap) {
    if (aux.negatives) return fail(ctx, RSI_ERR_UNSUPPORTED, "negative depth values");
    if (vm.inrange + aux.big != (uint64_t)n) return fail(ctx, RSI_ERR_INTERNAL, "value histogram does not add up to n");
    rsih::Quantiles q;   // hist_quantiles_int's result for the median (hostmath.h), from the device's walk
    q.med = (double)vm.lo;
    if (vm.lo <= vm.hi && (double)vm.hi - (double)vm.lo >= 1.0 && vm.med >= 0) q.med = (double)vm.med;
    if ((uint64_t)n / 2 > vm.inrange) {
      // the median lies above the histogram's 65 536 values (loaddata.cpp:233 takes it from the whole rescaled array): from the
      // array itself -- the rescaled int32 array the deep-coverage pass left, or the raw depth under -NOGC
      Phase ph_m(ctx, "a4.median on the host (depth > 65535)");
      if (P.gcadjust) { const int rcm = materialize_rd_gc(ctx); if (rcm != RSI_OK) return rcm; }
      std::vector<int32_t> all;
      const int rcf = fetch_i32(ctx, P.gcadjust ? ctx->rd_gc.as<int32_t>() : d_depth, n, all);
      if (rcf != RSI_OK) return rcf;
      q.med = rank_value_i32(all, (uint64_t)n / 2);
    }
    S.cap_median = q.med;
    capval = (int32_t)(q.med * P.cap);   // RD[i] = RDmedian*cap, truncated (loaddata.cpp:238)
  }

  ph_cap.stop();
  Phase ph_bins(ctx, "a5-9.compact+bins+stats");
  // ---- A5-A9: cap + compaction + bins + statistics (K4) ----
  std::vector<int64_t> cbreak(noncode.size()), cum(noncode.size() + 1, 0);
  for (size_t k = 0; k < noncode.size(); ++k) {
    cbreak[k] = (int64_t)noncode[k].start - cum[k];
    cum[k + 1] = cum[k] + (noncode[k].end - noncode[k].start + 1);
  }
  const int64_t ncompact = n - cum.back();
  const int64_t nb = ncompact / P.m;
  S.n_compact = ncompact; S.nbins = nb;
  ctx->ncompact = ncompact; ctx->nb = nb;
  if (ncompact <= 0 || nb < 8) return fail(ctx, RSI_ERR_TOO_SMALL, "nothing left after removing N regions");
  K4Regions inl;
  memset(&inl, 0, sizeof(inl));
  if ((int)noncode.size() <= kRegInline) {   // the usual case: the list rides with the kernel arguments
    for (size_t k = 0; k < cbreak.size(); ++k) inl.brk[k] = cbreak[k];
    for (size_t k = 0; k < cum.size(); ++k) inl.cum[k] = cum[k];
  } else {   // cbreak[4100] | cum: one upload
    std::vector<int64_t> both((size_t)4100 + cum.size(), 0);
    std::copy(cbreak.begin(), cbreak.end(), both.begin());
    std::copy(cum.begin(), cum.end(), both.begin() + 4100);
    HIPCHK(copy_h2d(ctx, d_cbreak, both.data(), both.size() * 8));
  }
  HIPCHK(ctx->rdc.ensure((size_t)(ncompact + 4) * 4));
  HIPCHK(ctx->binmed.ensure((size_t)nb * 4));
  HIPCHK(ctx->binsum.ensure((size_t)nb * 8));
  const size_t res_vals = want_cap && capval < kHistValues - 1 ? (size_t)std::max(capval, 0) + 1 : (size_t)kHistValues;
  BinAccum* d_bacc = reinterpret_cast<BinAccum*>(ctx->hist_res.p);
  uint32_t* d_res = reinterpret_cast<uint32_t*>(static_cast<char*>(ctx->hist_res.p) + kResHead);
  // K4's last workgroup folds the per-workgroup histograms and writes [BinAccum | histogram] into the mailbox.  With the
  // cap below the kernel's LDS value range the fold overwrites res_hist (nothing to clear); otherwise (no cap, or a cap
  // of 256 and more) stray values reach res_hist through global atomics and it is cleared first.
  size_t exp_bytes = kResHead + res_vals * kResClasses * 4;
  // the K4j that was queued behind K2j: accepted when it ran (its export replaced the marker) under exactly the regions, length
  // and cap the host has just derived itself, with nothing about the chromosome that the ordinary path treats differently
  PhaseParams hpp;
  memcpy(&hpp, head + kOffPhase, sizeof(hpp));
  bool spec_done = false;
  if (spec_slot) {
    int vr_g = 64, vr_c = 64;
    while (vr_g < 256 && vr_g <= ctx->spec_capval) vr_g <<= 1;
    while (vr_c < 256 && vr_c <= capval) vr_c <<= 1;
    const bool ran = spec_slot[3] != kSpecMagic;
    spec_done = ran && (P.gcadjust ? (joint_ok && !deep && !jinfo.esc_pending) : spec_nogc) && hpp.regions_ok == 1 && hpp.capval == capval && hpp.nreg == (int32_t)noncode.size() &&
                hpp.ncompact == ncompact && cap_compact8_applies(P.m, capval) && vr_g == vr_c && (ctx->spec_capval <= 127) == (capval <= 127) &&
                (int)noncode.size() <= kMaxRegions;
    if (ran && !spec_done) HIPCHK(hipMemsetAsync(ctx->hist_res.p, 0, kResHead + (size_t)256 * kResClasses * 4, st));   // it added its histogram to the rows: clean again
    ctx->phases.push_back({spec_done ? "spec.k4j accepted" : "spec.k4j rejected", 1.0});
  }
  uint32_t* exp_slot = spec_done ? spec_slot : (exp_bytes <= kMailboxMaxCopy ? static_cast<uint32_t*>(mb_alloc(ctx, exp_bytes)) : nullptr);
  if (spec_done) {
    ctx->rdc_is_bytes = true;   // all done by the queued launch
    if (!P.gcadjust) ctx->phases.push_back({"a5.nogc byte path", 1.0});
  } else if (P.gcadjust && want_cap && !deep && cap_compact8_applies(P.m, capval)) {
    // K4': from the byte copy of the raw depth, rescaling on the way -- the rescaled int32 array is never written or read
    HIPCHK(ctx->slabs.ensure(cap_compact8_slab_bytes(P.m, capval, ncompact)));
    HIPCHK(ctx->rdc8.ensure((size_t)ncompact + 64));
    ctx->rdc_is_bytes = true;
    // K2j's verified fixed-point ratios when it ran to the end; the float form with its exactness margin for a chromosome that
    // went through K2 + K3' (wrapped pair counters) -- RSI_HOT_K4J_FIX=0 selects that form for any chromosome
    const bool k4j_fix = joint_ok && !k4j_fix_off;
    const bool k4j = joint_ok || !(joint_env && atoi(joint_env) == 0);
    if (k4j && !k4j_fix) ctx->phases.push_back({"k4j.float rescale", 1.0});
    const bool split = k4j && k4j_fix && k4_split && rescale_compact_split_applies(P.m, capval, ncompact, (int)noncode.size());
    if (split) {
      HIPCHK(ctx->slabs.ensure(rescale_compact_split_slab_bytes(capval, ncompact)));
      HIPCHK(ctx->rdc8.ensure(rescale_compact_split_rdc_bytes(ncompact)));
      ctx->phases.push_back({"k4.split", 1.0});
    }
    Timer t(ctx, "cap_compact_bin", true);
    if (split) {   // K4s, K4m
      launch_rescale_compact_stream(ctx->depth8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, d_cbreak, d_cum, inl, (int)noncode.size(),
                                    ncompact, capval, P.m, ctx->rdc8.as<uint8_t>(), d_res, ctx->slabs.p, d_done + 2 * kDoneStride, ctx->hist_res.p, exp_slot,
                                    exp_slot ? exp_bytes : 0, reinterpret_cast<const unsigned int*>(ctx->joint_tot.as<uint8_t>() + joint_lut_off),
                                    &d_acc->escapes, nullptr, st);
      t.~Timer();   // (closes the streaming half's bracket: the medians have their own)
      new (&t) Timer(ctx, "bin_median", true);
      launch_bin_median8(ctx->rdc8.as<uint8_t>(), ncompact, capval, P.m, ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), nullptr, st);
    }
    else if (k4j)   // K4j: from the byte copy of the RAW depth (K2 and K2j both leave it), rescaling on the way
      launch_rescale_compact_bin8(ctx->depth8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, d_cbreak, d_cum, inl, (int)noncode.size(),
                                  ncompact, capval, P.m, ctx->rdc8.as<uint8_t>(), ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), d_res, ctx->slabs.p,
                                  ctx->gsum.p, d_done + 2 * kDoneStride, ctx->hist_res.p, exp_slot, exp_slot ? exp_bytes : 0,
                                  k4j_fix ? reinterpret_cast<const unsigned int*>(ctx->joint_tot.as<uint8_t>() + joint_lut_off) : nullptr, nullptr, st);
    else   // RSI_HOT_JOINT=0: round 2's chain to the end (K4' from K3''s rescaled bytes), kept for A/B runs
    launch_cap_compact_bin8(ctx->rescaled8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, d_cbreak, d_cum, inl, (int)noncode.size(),
                            ncompact, capval, P.m, ctx->rdc8.as<uint8_t>(), ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), d_res, ctx->slabs.p,
                            ctx->gsum.p, d_done + 2 * kDoneStride, ctx->hist_res.p, exp_slot, exp_slot ? exp_bytes : 0, st);
  } else if (!P.gcadjust && want_cap && cap_compact8_applies(P.m, capval) && !(getenv("RSI_HOT_NOGC_BYTES") && atoi(getenv("RSI_HOT_NOGC_BYTES")) == 0)) {
    // -NOGC with a cap below 254: K4' from the byte copy the histogram pass left (raw depth: no rescale anywhere)
    HIPCHK(ctx->slabs.ensure(cap_compact8_slab_bytes(P.m, capval, ncompact)));
    HIPCHK(ctx->rdc8.ensure((size_t)ncompact + 64));
    ctx->rdc_is_bytes = true;
    ctx->phases.push_back({"a5.nogc byte path", 1.0});
    const bool split = k4_split && rescale_compact_split_applies(P.m, capval, ncompact, (int)noncode.size());
    if (split) {
      HIPCHK(ctx->slabs.ensure(rescale_compact_split_slab_bytes(capval, ncompact)));
      HIPCHK(ctx->rdc8.ensure(rescale_compact_split_rdc_bytes(ncompact)));
      ctx->phases.push_back({"k4.split", 1.0});
    }
    Timer t(ctx, "cap_compact_bin", true);
    if (split) {   // K4s without ratios (the bytes are the values), K4m
      launch_rescale_compact_stream(ctx->rescaled8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, nullptr, d_cbreak, d_cum, inl, (int)noncode.size(),
                                    ncompact, capval, P.m, ctx->rdc8.as<uint8_t>(), d_res, ctx->slabs.p, d_done + 2 * kDoneStride, ctx->hist_res.p, exp_slot,
                                    exp_slot ? exp_bytes : 0, nullptr, nullptr, nullptr, st);
      t.~Timer();
      new (&t) Timer(ctx, "bin_median", true);
      launch_bin_median8(ctx->rdc8.as<uint8_t>(), ncompact, capval, P.m, ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), nullptr, st);
    } else
    launch_cap_compact_bin8(ctx->rescaled8.as<uint8_t>(), d_depth, ctx->gcbits.as<uint64_t>(), n, d_table, d_cbreak, d_cum, inl, (int)noncode.size(),
                            ncompact, capval, P.m, ctx->rdc8.as<uint8_t>(), ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), d_res, ctx->slabs.p,
                            ctx->gsum.p, d_done + 2 * kDoneStride, ctx->hist_res.p, exp_slot, exp_slot ? exp_bytes : 0, st, 1);
  } else {
    if (P.gcadjust) {   // no cap, a cap of 255 and more, or a wide bin: K4 from the rescaled int32 array, built first
      int rcm = materialize_rd_gc(ctx);
      if (rcm != RSI_OK) return rcm;
      d_src = ctx->rd_gc.as<int32_t>();
    }
    const int vbase = want_cap ? hist_window_base(S.cap_median, kK4Window) : (P.gcadjust ? hist_window_base(S.gc_rdmean, kK4Window) : 0);
    const char* w16_env = getenv("RSI_HOT_K4W");
    if (want_cap && cap_compact16_applies(P.m, capval, ncompact) && !(w16_env && atoi(w16_env) == 0)) {
      // K4w: a cap of 254 .. 32766 (deep coverage, or a generous cap): 16-bit tile and window counters.  RSI_HOT_K4W=0: the int32 kernel
      HIPCHK(hipMemsetAsync(d_res, 0, res_vals * kResClasses * 4, st));
      HIPCHK(ctx->slabs.ensure(cap_compact16_slab_bytes(P.m, ncompact)));
      ctx->phases.push_back({"a5.k4w 16-bit tile", 1.0});
      Timer t(ctx, "cap_compact_bin", true);
      launch_cap_compact_bin16(d_src, d_cbreak, d_cum, inl, (int)noncode.size(), ncompact, capval, P.m, ctx->rdc.as<int32_t>(), ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), d_res, ctx->slabs.p, d_done + 2 * kDoneStride, ctx->hist_res.p, exp_slot, exp_slot ? exp_bytes : 0, vbase, st);
    } else {
      const bool overwrite = cap_compact_overwrites(P.m, capval, ncompact, vbase) != 0;
      if (!overwrite) HIPCHK(hipMemsetAsync(d_res, 0, res_vals * kResClasses * 4, st));
      HIPCHK(ctx->slabs.ensure(cap_compact_slab_bytes(P.m, capval, ncompact, vbase)));
      Timer t(ctx, "cap_compact_bin", true);
      launch_cap_compact_bin(d_src, n, d_cbreak, d_cum, inl, (int)noncode.size(), ncompact, capval, P.m, ctx->rdc.as<int32_t>(), ctx->binmed.as<int32_t>(), ctx->binsum.as<int64_t>(), d_res, d_bacc, ctx->slabs.p, ctx->gsum.p, d_done + 2 * kDoneStride, ctx->hist_res.p, exp_slot, exp_slot ? exp_bytes : 0, vbase, st);
    }
  }
  BinAccum bacc;
  std::vector<uint32_t> hres_all(kResHead / 4 + res_vals * kResClasses);
  if (!exp_slot) HIPCHK(copy_d2h(ctx, hres_all.data(), ctx->hist_res.p, hres_all.size() * 4));
  if (!spec_done) HIPCHK(CTX_SYNC());
  if ((joint_ok || nogc_bytes) && want_cap && cap_compact8_applies(P.m, capval)) { ctx->spec_capval = capval; ctx->spec_m = P.m; ctx->spec_cap = (double)P.cap; ctx->spec_gc = P.gcadjust != 0; }
  if (exp_slot) memcpy(hres_all.data(), exp_slot, exp_bytes);
  memcpy(&bacc, hres_all.data(), sizeof(bacc));
  const uint32_t* hres = hres_all.data() + kResHead / 4;
  if (bacc.big) {
    // Compacted depths of 65 536 and more (no cap, or a cap above that: a median depth beyond 16 000 at -cap 4) are not in the
    // residue-class histogram.  The statistics it serves -- chromosome median and SD (rsi.cpp:2202-2203), the 31 subsamples' MADs
    // (rsi.cpp:1127-1143) -- come from the compacted array itself, on the host.
    Phase ph_h(ctx, "a6.statistics on the host (depth > 65535)");
    if (ctx->rdc_is_bytes) return fail(ctx, RSI_ERR_INTERNAL, "values beyond the histogram on the byte path");
    std::vector<int32_t> rd;
    const int rcf = fetch_i32(ctx, ctx->rdc.as<int32_t>(), ncompact, rd);
    if (rcf != RSI_OK) return rcf;
    S.RDsd = sqrt(rsih::variance_pop(rd.data(), (size_t)ncompact));   // the reference's own loop: double sums in index order (wufunctions.cpp:766-809)
    const uint64_t sublen = (uint64_t)(ncompact / 31);
    std::vector<int32_t> sub((size_t)sublen);
    std::vector<int32_t> sorted = rd;
    const double RDmed = rank_value_i32(sorted, (uint64_t)ncompact / 2);
    for (int j = 0; j < 31 && sublen > 0; ++j) {
      for (uint64_t k = 0; k < sublen; ++k) sub[(size_t)k] = (int)fabs((float)rd[(size_t)(j + 31 * k)] - RDmed);   // rsi.cpp:1134
      pb.mads[j] = rank_value_i32(sub, sublen / 2);
    }
    S.RDmedian = RDmed;
    pb.host_stats = true;
    pb.ncompact = ncompact; pb.nb = nb; pb.res_vals = 0; pb.RDmedian = RDmed;
    return RSI_OK;
  }
  // chromosome median / SD (rsi.cpp:2202-2203)
  std::vector<uint64_t> hall(res_vals, 0);
  for (size_t v = 0; v < res_vals; ++v) for (int c = 0; c < kResClasses; ++c) hall[v] += hres[v * kResClasses + c];
  rsih::Quantiles qall;
  if (!int_quantiles(hall, (uint64_t)ncompact, qall)) return fail(ctx, RSI_ERR_INTERNAL, "empty depth histogram");
  const double RDmedian = qall.med;
  {   // variance(RD,...,-1), wufunctions.cpp:766-809: exact integer sums, one rounding each
    unsigned __int128 s1 = 0, s2 = 0;
    for (size_t v = 0; v < res_vals; ++v) { s1 += (unsigned __int128)v * hall[v]; s2 += (unsigned __int128)v * v * hall[v]; }
    const double d1 = (double)s1, d2 = (double)s2;
    const double mean = d1 / double((int)ncompact);
    S.RDsd = sqrt(d2 / double((int)ncompact) - mean * mean);
  }
  S.RDmedian = RDmedian;
  pb.ncompact = ncompact; pb.nb = nb; pb.res_vals = res_vals; pb.RDmedian = RDmedian;
  pb.hres_all = std::move(hres_all);
  return RSI_OK;
}

// A9-A19 on the bins and the candidates: MAD, NB transform, the two scans, segments, block tests, candidate stages.
int bin_level_stages(rsi_ctx* ctx, const rsi_params& P, int64_t n, rsi_result* res, const PerBase& pb,
                     std::vector<Candidate>& blocks, std::vector<Candidate>& raw, std::vector<Candidate>& kept,
                     std::vector<Candidate>& segs_all) {
  rsi_chrom_stats& S = res->stats;
  uint8_t* small = ctx->small.as<uint8_t>();
  hipStream_t st = ctx->stream;
  const std::vector<Region>& noncode = pb.noncode;
  const int64_t ncompact = pb.ncompact, nb = pb.nb;
  const size_t res_vals = pb.res_vals;
  const uint32_t* hres = pb.hres_all.data() + kResHead / 4;
  const double RDmedian = pb.RDmedian;
  int rc = RSI_OK;

  if (!(RDmedian < 5)) {   // rsi.cpp:1809-1812
    Phase ph_nb(ctx, "a9-10.mad+nb");
    GateShared gs_nb(ctx);
    // ---- A9: MAD of the 31 interleaved subsamples from their value histograms (rsi.cpp:1127-1143) ----
    double mads[31];
    const uint64_t sublen = (uint64_t)(ncompact / 31);
    if (sublen == 0) return fail(ctx, RSI_ERR_TOO_SMALL, "fewer than 31 bases");
    for (int j = 0; j < 31; ++j) {
      if (pb.host_stats) { mads[j] = pb.mads[j]; continue; }
      std::vector<uint64_t> hd(res_vals + 1, 0);
      for (size_t v = 0; v < res_vals; ++v) {
        const uint32_t c = hres[v * kResClasses + j];
        if (!c) continue;
        const int a = (int)fabs((float)(int)v - RDmedian);    // RDtmp[k]=abs((float)RD[i]-RDmedian), rsi.cpp:1134
        hd[(size_t)a] += c;
      }
      rsih::Quantiles qd;
      if (!int_quantiles(hd, sublen, qd)) return fail(ctx, RSI_ERR_INTERNAL, "empty MAD histogram");
      mads[j] = qd.med;
    }
    const double mad = rsih::grid_quantiles(mads, (size_t)31).med;
    const double r = RDmedian / mad;
    S.nb_mad = mad; S.nb_r = r;
    const double factor = sqrt(2.0 * (1.0 + P.epsilon) * log(3.1E9));   // rsi.cpp:1829
    const int LmaxBase = std::max(20, 10000 / P.m);                      // rsi.cpp:1830-1831

    HIPCHK(ctx->first_del.ensure((size_t)(2 * nb + 8) * 4));   // first_del | first_dup, set to "no L" by one memset
    HIPCHK(ctx->status1.ensure((size_t)nb * 4));
    HIPCHK(ctx->status1f.ensure((size_t)nb * 4));
    HIPCHK(ctx->status2.ensure((size_t)nb * 4));
    HIPCHK(ctx->runs.ensure((size_t)kMaxRunEntries * 8));
    HIPCHK(ctx->scan_tiles.ensure((size_t)(nb / 256 + 2) * 4));   // the scan's list of tiles that can hit (launch_rsi_scan)
    HIPCHK(ctx->fs_ws.ensure(level_sums_workspace_bytes(nb, kFsListCap, kMaxL)));
    HIPCHK(ctx->fs_out.ensure((size_t)(2 * kMaxL + 1) * 8 + 64));

    // ---- A10: NB transform (K5), always computed as the reference does (Q10).  The raw minimum stays on the device: the
    // scaling kernel derives the scaled levels from it and from the three raw reference levels computed here (host libm, as
    // the reference), and takes the min/max of the scaled values for the median that follows. ----
    HIPCHK(ctx->tnb.ensure((size_t)nb * 4));
    HIPCHK(ctx->hist_f.ensure((size_t)kGridCap * 4));
    uint32_t* d_rawmin = reinterpret_cast<uint32_t*>(small + kOffRawMin);
    GridMedian* d_g = reinterpret_cast<GridMedian*>(small + kOffGrid);
    { Timer t(ctx, "nb_raw"); launch_nb_raw(ctx->binsum.as<int64_t>(), nb, P.m, ncompact, r, ctx->tnb.as<float>(), d_rawmin, st); }
    auto nbf = [&](double sum) {
      const double mm = (double)P.m;
      return 2.0 * sqrt(r) * log(sqrt((sum + 0.25) / (mm * r - 0.5)) + sqrt(1.0 + (sum + 0.25) / (mm * r - 0.5)));
    };
    const double med_raw = nbf(RDmedian * P.m), del_raw = nbf(RDmedian / 2.0 * (double)P.m), dup_raw = nbf(RDmedian * 1.5 * (double)P.m);
    { Timer t(ctx, "nb_scale"); launch_nb_scale_minmax(ctx->tnb.as<float>(), nb, d_rawmin, med_raw, del_raw, dup_raw, RDmedian, grid_chain(ctx), d_g, st); }
    bool nb_planned = true;   // d_g[0] holds the grid of the NB values' median, its buckets are clear
    ctx->have_nb = true;

    ph_nb.stop();
    gs_nb.release();
    rsih::CallerInput in;
    int short_neighbourhoods = 0;
    in.short_neighbourhoods = &short_neighbourhoods;
    in.P = P; in.RDmedian = RDmedian; in.RDsd = S.RDsd; in.ncompact = ncompact; in.noncode = &noncode;
    HIPCHK(ctx->h_medint.ensure((size_t)nb * 4));
    // The bin medians for the host's block tests: K4 is complete (the host has its statistics), so the copy needs no ordering
    // against the stream -- it goes to a stream of its own and runs beside the transform, the quantile chains and the scan
    // instead of in front of them (a blit of 30 - 170 us at the head of a lone chromosome's chain).  Joined before the block
    // tests, and at the latest when this run ends (CopyJoin): the next run's K4 writes the same array.
    // (the guard exists before the copy is queued, and the copy counts as pending from the moment it is: every return joins it;
    // the join is a wait with a deadline that poisons the context like any other, pipeline_internal.h: join_copy)
    struct CopyJoin { rsi_ctx* c; ~CopyJoin() { (void)join_copy(c); } } copy_join{ctx};
    HIPCHK(hipMemcpyAsync(ctx->h_medint.p, ctx->binmed.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->copy_stream));
    ctx->copy_pending = true;
    HIPCHK(hipEventRecord(ctx->copy_ev, ctx->copy_stream));
    ctx->copy_recorded = true;
    in.binmedint = rsih::IntSpan(ctx->h_medint.as<int>(), nb);   // read in place, after join_copy()

    auto do_scan = [&](bool use_med, std::vector<Candidate>& segs) -> int {
      ScanOut so;
      const float* d_T;
      ChainOut* first = nullptr;
      int rc2;
      if (use_med) {
        HIPCHK(ctx->tmed.ensure((size_t)nb * 4));
        { Timer t(ctx, "i32_to_f32"); launch_i32_to_f32_minmax(ctx->binmed.as<int32_t>(), ctx->tmed.as<float>(), nb, RDmedian, grid_chain(ctx), d_g + 1, st); }
        d_T = ctx->tmed.as<float>();
        ctx->have_med = true;
        nb_planned = false;   // the chains share the bucket array
        if ((rc2 = grid_pair_issue(ctx, d_T, nullptr, nb, false, RDmedian, true, 0, &first)) != RSI_OK) return rc2;
      } else {
        d_T = ctx->tnb.as<float>();
        if ((rc2 = grid_pair_issue(ctx, d_T, nullptr, nb, true, 0.0, nb_planned, 0, &first)) != RSI_OK) return rc2;
        nb_planned = false;
      }
      rc2 = run_scan(ctx, P, use_med, d_T, nb, RDmedian, factor, LmaxBase, first, med_raw, del_raw, dup_raw, so);
      if (rc2 != RSI_OK) return rc2;
      {
        const uint32_t key = ~first->rawmin_inv;
        float tminf;
        { const uint32_t bb = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key; memcpy(&tminf, &bb, 4); }
        S.nb_tmin = tminf;
      }
      ctx->last_scan_med = use_med;
      S.tmedian1 = so.tmedian1; S.tsigma1 = so.tsigma1; S.tlamda1 = so.tlamda1;
      S.tmedian2 = so.tmedian2; S.tsigma2 = so.tsigma2; S.tlamda2 = so.tlamda2;
      S.Lmax = so.Lmax; S.trim_escapes += (int)so.escapes; S.inexact_sums = (int)so.inexact;
      S.scan_tiles = (int32_t)((nb + 255) / 256); S.scan_tiles_listed = (int32_t)so.tiles_listed;
      for (int w = 0; w < 4; ++w) { res->level_log[w] = so.level_log[w]; res->stop_levels[w] = so.stop_levels[w]; }
      res->fs_lines = so.fs_lines;
      res->log_nb = !use_med;
      for (const Candidate& c : so.segs) segs_all.push_back(c);
      segs = so.segs;
      {   // areblockscnv, rsi.cpp:1847: on the bin medians; a scan with many segments sends its first round to the device as one batch
        Phase ph(ctx, "a15.blocks");
        HIPCHK(join_copy(ctx));   // the host copy of the bin medians
        const char* bb_env = getenv("RSI_HOT_BLOCK_BATCH");   // 0: every block test on the host
        DeviceTester block_tester(ctx, DepthRef{ctx->binmed.p, 4}, nb, RDmedian);
        rsih::CallProfile bprof;
        in.block_tester = (bb_env && atoi(bb_env) == 0) ? nullptr : &block_tester;
        in.prof = &bprof;
        rsih::test_block_segments(in, so.status2, segs);
        in.block_tester = nullptr;
        in.prof = nullptr;
        if (block_tester.failed) return RSI_ERR_HIP;
        if (bprof.block_batch_hits) ctx->phases.push_back({"a15.block batch hits", (double)bprof.block_batch_hits});
      }
      return RSI_OK;
    };
    std::vector<Candidate> tested;
    if (P.trans != 0) { if ((rc = do_scan(true, tested)) != RSI_OK) return rc; }          // rsi.cpp:1837-1840
    if (P.trans == 0) { if ((rc = do_scan(false, tested)) != RSI_OK) return rc; }         // rsi.cpp:1845-1849
    if (P.trans == 2) {                                                                   // rsi.cpp:1852-1858
      std::vector<Candidate> more;
      if ((rc = do_scan(false, more)) != RSI_OK) return rc;
      tested.insert(tested.end(), more.begin(), more.end());
    }
    auto mirror_source = [ctx, ncompact]() -> int32_t* {   // grow-only pinned host mirror: transfers land in it directly
      if (materialize_rdc(ctx) != RSI_OK) return nullptr;   // the pages are int32: widen the byte array first (same stream as the copies)
      if (ctx->mirror_cap < (size_t)ncompact) {
        if (ctx->mirror) (void)hipHostFree(ctx->mirror);
        ctx->mirror = nullptr;
        ctx->mirror_cap = (size_t)ncompact + (size_t)ncompact / 8 + 1024;
        if (hipHostMalloc(reinterpret_cast<void**>(&ctx->mirror), ctx->mirror_cap * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
          ctx->mirror = nullptr; ctx->mirror_cap = 0;
        }
      }
      return ctx->mirror;
    };
    rsih::DepthPager pager(ctx->rdc.as<int32_t>(), ncompact, st, mirror_source, nullptr, 0, ctx->sync_ev);
    rsih::CallProfile prof;
    in.prof = &prof;
    const DepthRef depth_ref = ctx->rdc_is_bytes ? DepthRef{ctx->rdc8.p, 1} : DepthRef{ctx->rdc.p, 4};
    DeviceTester tester(ctx, depth_ref, ncompact, RDmedian);
    const char* host_env = getenv("RSI_HOT_HOST_CANDIDATES");   // debugging switch: candidate stages on the host
    const bool host_tests = host_env && atoi(host_env) != 0;
    in.tester = host_tests ? nullptr : &tester;
    { Phase ph(ctx, "a16-19.calls"); rsih::call_from_segments(in, tested, pager, blocks, raw, kept); }
    if (tester.failed) return RSI_ERR_HIP;
    if (short_neighbourhoods) return fail(ctx, RSI_ERR_UNSUPPORTED, "a candidate longer than the neighbourhood left around it (the reference aborts in isitcnv, rsi.cpp:107)");
    if (pager.failed()) return fail(ctx, RSI_ERR_INTERNAL, "no pinned host memory for the candidate stages' depth mirror");
    ctx->phases.push_back({"calls.device(wait)", tester.kernel_wait_ms});
    ctx->phases.push_back({"calls.device_ms", prof.device_ms});
    ctx->phases.push_back({"calls.final", prof.final_tests});
    ctx->phases.push_back({"calls.launches", (double)tester.launches});
    ctx->phases.push_back({"calls.spec_hits", (double)prof.spec_hits});
    ctx->phases.push_back({"calls.single_tests", (double)prof.single_tests});
    ctx->phases.push_back({"calls.host_fallbacks", (double)prof.host_fallbacks});
    ctx->phases.push_back({"calls.fetch", pager.fetch_ms()});
    ctx->phases.push_back({"calls.gather(incl fetch)", prof.gather});
    ctx->phases.push_back({"calls.winmean", prof.winmean});
    ctx->phases.push_back({"calls.quantiles", prof.quantiles});
    ctx->phases.push_back({"calls.variance", prof.variance});
    ctx->phases.push_back({"calls.sharpen", prof.sharpen});
    ctx->phases.push_back({"calls.merge(incl tests)", prof.merge});
    ctx->phases.push_back({"calls.ntests", (double)prof.tests});
  }
  return RSI_OK;
}

int run_device_impl(rsi_ctx* ctx, const rsi_params* Pp, const int32_t* d_depth, const uint8_t* d_fasta, int64_t n,
                    rsi_result* res) {
  const rsi_params& P = *Pp;
  const double t_begin = now_ms();
  if (n <= 0 || n >= (1ll << 31) - 4096) return fail(ctx, RSI_ERR_BAD_ARG, "chromosome length must be in (0, 2^31)");
  if (P.m < 1 || (P.m % 2) != 1) return fail(ctx, RSI_ERR_BAD_ARG, "m must be odd (the reference forces it, rsi.cpp:2061-2064)");
  if (P.m > 3000) return fail(ctx, RSI_ERR_UNSUPPORTED, "bin size above 3000 is not supported by the bin kernel");
  if (((uintptr_t)d_depth & 15) || ((uintptr_t)d_fasta & 15)) return fail(ctx, RSI_ERR_BAD_ARG, "device inputs must be 16-byte aligned");
  if (P.gcadjust && n / 20 <= 201)
    return fail(ctx, RSI_ERR_TOO_SMALL, "size of RDA should be much larger than bin size (gccontent.cpp:66-71)");
  HIPCHK(hipSetDevice(ctx->device));
  ctx->ktimes.clear();
  ctx->event_next = 0;
  ctx->phases.clear();
  { const int64_t reserve = ctx->reserve_n.load(); tl_grow = reserve > n ? (double)reserve / (double)n : 1.0; }
  tl_grow_ms = 0.0;
  if (!ctx_enter(ctx)) return RSI_ERR_HIP;
  mailbox_reset(ctx);
  PerBase pb;
  int rc = per_base_phase(ctx, P, d_depth, d_fasta, n, res, pb);
  if (rc != RSI_OK) return rc;
  rsi_chrom_stats& S = res->stats;
  std::vector<Candidate> blocks, raw, kept, segs_all;
  if ((rc = bin_level_stages(ctx, P, n, res, pb, blocks, raw, kept, segs_all)) != RSI_OK) return rc;
  Phase ph_fin(ctx, "z.finish");
  const std::vector<Candidate>* lists[4] = {&kept, &raw, &segs_all, &blocks};
  for (int w = 0; w < 4; ++w) {
    res->lists[w].resize(lists[w]->size());
    for (size_t i = 0; i < lists[w]->size(); ++i) to_call((*lists[w])[i], &res->lists[w][i]);
  }
  HIPCHK(CTX_SYNC());
  ph_fin.stop();
  ctx->phases.push_back({"mem.grow", tl_grow_ms});
  S.t_device_ms = now_ms() - t_begin;
  ctx->phases.push_back({"z.total", S.t_device_ms});
  if (ctx->timing) {
    double tot = 0;
    for (const KernelTime& k : ctx->ktimes) { float ms = 0; (void)hipEventElapsedTime(&ms, k.a, k.b); tot += ms; }
    S.t_kernels_ms = tot;
  }
  return RSI_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" {

void rsi_default_params(rsi_params* p) {   // rsi.cpp:34-98
  p->m = 101; p->gcadjust = 1; p->trans = 0; p->merge = 1; p->maxchkbp = 100000; p->debug = 0;
  p->cap = 4.0; p->epsilon = 1.5; p->threshold = -1.0; p->chklen = 2.5; p->minmlen = 3.01; p->buffer = 0.05; p->p = 0.05;
}

// One side stream per DEVICE for the contexts' small device -> host copies that need no ordering against their kernels (the bin
// medians for the block tests): a stream per context doubled a 16-worker pool's streams beyond the hardware queues, and the genome
// took 45 ms instead of 13.  The copies of different chromosomes queue behind each other here (1.6 ms of copy time per genome).
static hipStream_t shared_copy_stream(int device) {
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!streams[device] && hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) streams[device] = nullptr;
  return streams[device];
}

rsi_ctx* rsi_hot_create(int device, int* status) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0 || device < 0 || device >= count) {
    set_global_error("no usable HIP device (librsi_hot has no CPU fallback)");
    if (status) *status = RSI_ERR_NO_DEVICE;
    return nullptr;
  }
  rsi_ctx* ctx = new rsi_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->sync_ev, hipEventDisableTiming) != hipSuccess ||
      (ctx->copy_stream = shared_copy_stream(device)) == nullptr ||
      hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming) != hipSuccess) {
    set_global_error("hipSetDevice / hipStreamCreate failed");
    if (status) *status = RSI_ERR_HIP;
    delete ctx;
    return nullptr;
  }
  if (status) *status = RSI_OK;
  return ctx;
}

void rsi_hot_destroy(rsi_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->mirror) (void)hipHostFree(ctx->mirror);
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  for (int b = 0; b < 2; ++b) if (ctx->text_pin[b]) (void)hipHostFree(ctx->text_pin[b]);
  if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
  if (ctx->copy_ev) (void)hipEventDestroy(ctx->copy_ev);   // (copy_stream is the device's shared one: never destroyed)
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* rsi_hot_last_error(const rsi_ctx* ctx) {
  if (ctx) return ctx->err.c_str();
  static thread_local std::string copy;   // the global record changes under other threads' feet: hand out a snapshot
  { std::lock_guard<std::mutex> lk(g_err_mu); copy = g_last_error; }
  return copy.c_str();
}

void rsi_hot_set_timing(rsi_ctx* ctx, int on) { if (ctx) ctx->timing = on < 0 ? 0 : on > 3 ? 1 : on; }
void rsi_hot_set_timing_kernel(rsi_ctx* ctx, const char* name) { if (ctx) ctx->timing_kernel = (name && name[0]) ? name : "cap_compact_bin"; }

int rsi_hot_run_device(rsi_ctx* ctx, const rsi_params* p, const void* d_depth, const void* d_fasta, int64_t n, rsi_result** out) {
  if (!ctx || !p || !d_depth || !d_fasta || !out) return fail(ctx, RSI_ERR_BAD_ARG, "null argument");
  rsi_result* res = new rsi_result();
  int rc = run_device_impl(ctx, p, static_cast<const int32_t*>(d_depth), static_cast<const uint8_t*>(d_fasta), n, res);
  if (rc != RSI_OK) { delete res; *out = nullptr; return rc; }
  *out = res;
  return RSI_OK;
}


int rsi_hot_run(rsi_ctx* ctx, const rsi_params* p, const int32_t* depth, const uint8_t* fasta, int64_t n, rsi_result** out) {
  if (!ctx || !p || !depth || !fasta || !out) return fail(ctx, RSI_ERR_BAD_ARG, "null argument");
  if (n <= 0) return fail(ctx, RSI_ERR_BAD_ARG, "empty chromosome");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(ctx->in_depth.ensure((size_t)(n + 4) * 4));
  HIPCHK(ctx->in_fasta.ensure((size_t)n + 64));
  ctx->n_in = n;
  // The depth crosses PCIe as BYTES where that pays: the host narrows its int32 array (AVX2: ten gigabytes a second and core)
  // into pinned memory, the values of 255 and more -- next to none at sequencing depths -- travel as a list, the device widens
  // again (1 + 4 bytes per base of HBM traffic, 0.1 ms per 100 Mb): 2 bytes per base over the link instead of 5.  One thread
  // narrows slower than the link copies, so this is for a pool whose workers each bring a chromosome of their own (the link
  // is what sixteen of them share); a lone call copies the array as it is.  RSI_HOT_H2D_NARROW=0 / 1 forces either.
  const char* narrow_v = getenv("RSI_HOT_H2D_NARROW");
  const int narrow_env = narrow_v ? atoi(narrow_v) : -1;
  const bool narrow = n >= (1 << 18) && (narrow_env >= 0 ? narrow_env != 0 : (ctx->gate != nullptr && !ctx->gate->lonely()));
  bool narrowed = false;
  if (narrow) {
    Phase ph(ctx, "h2d.narrow");
    const int64_t cap = n / 64 + 16;
    const size_t list_off = ((size_t)n + 63) & ~size_t(63);
    HIPCHK(ctx->h_d8.ensure(list_off + (size_t)cap * 8));
    uint8_t* h8 = ctx->h_d8.as<uint8_t>();
    int32_t* hpos = reinterpret_cast<int32_t*>(h8 + list_off);
    int32_t* hval = hpos + cap;
    // a long chromosome's narrowing on two threads (its caller's and one more): at ten gigabytes a second the 1 GB of a 250 Mb
    // chromosome is what a pooled genome's makespan would otherwise end on
    int64_t nesc = 0;
    const char* split_v = getenv("RSI_HOT_H2D_SPLIT_MIN");   // (bases from which two threads narrow; tests lower it)
    if (n >= (split_v ? atoll(split_v) : (long long)96 << 20)) {
      const int64_t half = (n / 2) & ~(int64_t)63, cap0 = cap / 2, cap1 = cap - cap0;
      int64_t ne1 = 0;
      std::thread helper([&] { ne1 = rsih::narrow_depth_u8(depth + half, n - half, h8 + half, hpos + cap0, hval + cap0, cap1); });
      const int64_t ne0 = rsih::narrow_depth_u8(depth, half, h8, hpos, hval, cap0);
      helper.join();
      if (ne0 <= cap0 && ne1 <= cap1) {   // the second half's entries move up behind the first's, their positions counted from the array's start
        for (int64_t k = 0; k < ne1; ++k) { hpos[ne0 + k] = hpos[cap0 + k] + (int32_t)half; hval[ne0 + k] = hval[cap0 + k]; }
        nesc = ne0 + ne1;
      } else nesc = cap + 1;
    } else nesc = rsih::narrow_depth_u8(depth, n, h8, hpos, hval, cap);
    if (nesc <= cap) {
      HIPCHK(ctx->in_d8.ensure((size_t)n + 64));
      HIPCHK(hipMemcpyAsync(ctx->in_d8.p, h8, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
      launch_widen_u8(ctx->in_d8.as<uint8_t>(), n, ctx->in_depth.as<int32_t>(), ctx->stream);
      if (nesc > 0) {
        HIPCHK(ctx->in_esc.ensure((size_t)cap * 8));
        HIPCHK(hipMemcpyAsync(ctx->in_esc.p, hpos, (size_t)cap * 8, hipMemcpyHostToDevice, ctx->stream));
        launch_patch_i32(ctx->in_depth.as<int32_t>(), ctx->in_esc.as<int32_t>(), ctx->in_esc.as<int32_t>() + cap, nesc, ctx->stream);
      }
      narrowed = true;
    }
  }
  if (!narrowed) HIPCHK(copy_h2d(ctx, ctx->in_depth.p, depth, (size_t)n * 4));
  HIPCHK(copy_h2d(ctx, ctx->in_fasta.p, fasta, (size_t)n));
  HIPCHK(CTX_SYNC());
  return rsi_hot_run_device(ctx, p, ctx->in_depth.p, ctx->in_fasta.p, n, out);
}

int rsi_result_ncalls(const rsi_result* r, int which) { return (r && which >= 0 && which < 4) ? (int)r->lists[which].size() : 0; }
const rsi_call* rsi_result_calls(const rsi_result* r, int which) {
  return (r && which >= 0 && which < 4 && !r->lists[which].empty()) ? r->lists[which].data() : nullptr;
}
const rsi_chrom_stats* rsi_result_stats(const rsi_result* r) { return r ? &r->stats : nullptr; }
int rsi_result_noncode(const rsi_result* r, int32_t* pairs, int cap) {
  if (!r) return 0;
  const int k = (int)r->noncode.size() / 2;
  for (int i = 0; i < k && i < cap; ++i) { pairs[2 * i] = r->noncode[2 * i]; pairs[2 * i + 1] = r->noncode[2 * i + 1]; }
  return k;
}
void rsi_result_free(rsi_result* r) { delete r; }

int64_t rsi_hot_fetch_i32(rsi_ctx* ctx, const char* name, int32_t* out, int64_t cap) {
  if (!ctx || !name) return RSI_ERR_BAD_ARG;
  const std::string s(name);
  const void* src = nullptr; int64_t cnt = 0;
  if (s == "rd_gc" && ctx->have_gc) {
    if (out) { HIPCHK(hipSetDevice(ctx->device)); const int rcm = materialize_rd_gc(ctx); if (rcm != RSI_OK) return rcm; }
    src = out ? ctx->rd_gc.p : static_cast<const void*>(ctx); cnt = ctx->n;
  }
  else if (s == "rd_concat") {
    if (out) { HIPCHK(hipSetDevice(ctx->device)); const int rcm = materialize_rdc(ctx); if (rcm != RSI_OK) return rcm; }
    src = ctx->rdc.p ? ctx->rdc.p : static_cast<const void*>(ctx); cnt = ctx->ncompact;
  }
  else if (s == "depth_in" && ctx->in_depth.p) { src = ctx->in_depth.p; cnt = ctx->n_in; }
  else if (s == "binmedint") { src = ctx->binmed.p; cnt = ctx->nb; }
  else if (s == "status1") { src = ctx->status1.p; cnt = ctx->nb; }
  else if (s == "status1f") { src = ctx->status1f.p; cnt = ctx->nb; }
  else if (s == "status2") { src = ctx->status2.p; cnt = ctx->nb; }
  // diagnostics of the last run's per-base phase: K2j's fixed-point ratios per GC level (bit 31: not verified, bit 30: the level
  // does not occur)
  else if (s == "k4_ratios" && ctx->joint_tot.p) {
    const size_t list_off = (gc_joint_totals_bytes() + 255) & ~size_t(255), lut_off = list_off + ((gc_joint_esc_list_bytes() + 255) & ~size_t(255));
    src = ctx->joint_tot.as<uint8_t>() + lut_off; cnt = kGcLevels;
  }
  if (!src) return fail(ctx, RSI_ERR_BAD_ARG, "unknown or unavailable array: " + s);
  if (out) {
    const int64_t k = std::min(cnt, cap);
    if (!ctx_enter(ctx)) return RSI_ERR_HIP;
    mailbox_reset(ctx);
    HIPCHK(copy_d2h(ctx, out, src, (size_t)k * 4));
    HIPCHK(CTX_SYNC());
  }
  return cnt;
}
int64_t rsi_hot_fetch_f32(rsi_ctx* ctx, const char* name, float* out, int64_t cap) {
  if (!ctx || !name) return RSI_ERR_BAD_ARG;
  const std::string s(name);
  const void* src = nullptr; int64_t cnt = 0;
  if (s == "binnb" && ctx->have_nb) { src = ctx->tnb.p; cnt = ctx->nb; }
  else if (s == "binmed" && ctx->have_med) { src = ctx->tmed.p; cnt = ctx->nb; }
  if (!src) return fail(ctx, RSI_ERR_BAD_ARG, "unknown or unavailable array: " + s);
  if (out) {
    const int64_t k = std::min(cnt, cap);
    if (!ctx_enter(ctx)) return RSI_ERR_HIP;
    mailbox_reset(ctx);
    HIPCHK(copy_d2h(ctx, out, src, (size_t)k * 4));
    HIPCHK(CTX_SYNC());
  }
  return cnt;
}
int64_t rsi_hot_fetch_i64(rsi_ctx* ctx, const char* name, int64_t* out, int64_t cap) {
  if (!ctx || !name) return RSI_ERR_BAD_ARG;
  const std::string s(name);
  if (s != "binsum" || ctx->nb == 0) return fail(ctx, RSI_ERR_BAD_ARG, "unknown or unavailable array: " + s);
  if (out) {
    const int64_t k = std::min(ctx->nb, cap);
    if (!ctx_enter(ctx)) return RSI_ERR_HIP;
    mailbox_reset(ctx);
    HIPCHK(copy_d2h(ctx, out, ctx->binsum.p, (size_t)k * 8));
    HIPCHK(CTX_SYNC());
  }
  return ctx->nb;
}

// Test hook: filterstatus' per-level sums of host arrays through the device kernels (include/rsi_hot.h).
int rsi_hot_debug_level_sums(rsi_ctx* ctx, const float* T, const int32_t* status, int64_t nb, int Lmax, float* sums, int32_t* counts) {
  if (!ctx || !T || !status || !sums || !counts || nb <= 0 || Lmax < 1 || Lmax > kMaxL) return fail(ctx, RSI_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx_enter(ctx)) return RSI_ERR_HIP;
  mailbox_reset(ctx);
  HIPCHK(ctx->small.ensure(kSmallBytes));
  HIPCHK(ctx->tnb.ensure((size_t)nb * 4));
  HIPCHK(ctx->status1.ensure((size_t)nb * 4));
  HIPCHK(ctx->fs_ws.ensure(level_sums_workspace_bytes(nb, kFsListCap, kMaxL)));
  HIPCHK(ctx->fs_out.ensure((size_t)(2 * kMaxL + 1) * 8 + 64));
  HIPCHK(hipMemcpyAsync(ctx->tnb.p, T, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->status1.p, status, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
  unsigned int* counter = reinterpret_cast<unsigned int*>(ctx->small.as<uint8_t>() + kOffDone) + kDoneBinSlot + 4;
  HIPCHK(hipMemsetAsync(ctx->fs_ws.p, 0, level_sums_head_bytes(kMaxL), ctx->stream));
  HIPCHK(hipMemsetAsync(counter, 0, 4, ctx->stream));
  const int nlev = 2 * Lmax + 1;
  uint32_t* slot = static_cast<uint32_t*>(mb_alloc(ctx, (size_t)nlev * 8));
  if (!slot) return fail(ctx, RSI_ERR_INTERNAL, "out of pinned mailbox memory");
  { Timer t(ctx, "level_sums"); launch_level_sums(ctx->tnb.as<float>(), ctx->status1.as<int32_t>(), nb, Lmax, ctx->fs_ws.p, kMaxL, kFsListCap, ctx->fs_out.as<float>(), counter, slot, ctx->stream); }
  HIPCHK(CTX_SYNC());
  memcpy(sums, slot, (size_t)nlev * 4);
  memcpy(counts, slot + nlev, (size_t)nlev * 4);
  return RSI_OK;
}

// Test hook: one scan pass (rsistatus, rsi.cpp:1191-1259: detection, exact sweep, level stop, first-mark resolution) over host
// arrays -- the transformed bins, the integer bin medians, and the thresholds the caller chose.  For window lengths no whole
// run reaches in test time (-m 1: the stages behind the scan take the reference, and this library's host side, hours).
int rsi_hot_debug_scan(rsi_ctx* ctx, const float* T, const int32_t* medint, int64_t nb, double RDmedian, double tmedian, double tlamda,
                       int Lmax, int32_t* status, int32_t* info /* [4]: tiles listed, trim escapes, inexact, 0 */) {
  if (!ctx || !T || !medint || !status || nb <= 0 || nb >= (1ll << 31) - 4096 || Lmax < 1 || Lmax > kHardMaxL || Lmax > nb) return fail(ctx, RSI_ERR_BAD_ARG, "bad argument");
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx_enter(ctx)) return RSI_ERR_HIP;
  mailbox_reset(ctx);
  ctx->nb = nb;
  HIPCHK(ctx->small.ensure(kSmallBytes));
  HIPCHK(ctx->tnb.ensure((size_t)nb * 4));
  HIPCHK(ctx->binmed.ensure((size_t)nb * 4));
  HIPCHK(ctx->first_del.ensure((size_t)(2 * nb + 8) * 4));
  HIPCHK(ctx->status1.ensure((size_t)nb * 4));
  HIPCHK(ctx->status1f.ensure((size_t)nb * 4));
  HIPCHK(ctx->runs.ensure((size_t)kMaxRunEntries * 8));
  HIPCHK(ctx->scan_tiles.ensure((size_t)(nb / 256 + 2) * 4));
  HIPCHK(hipMemcpyAsync(ctx->tnb.p, T, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->binmed.p, medint, (size_t)nb * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->small.p, 0, kHeaderBytes, ctx->stream));   // arrival counters, list counts
  {
    FillList fl{};
    void* keep = ctx->fs_ws.p;
    ctx->fs_ws.p = nullptr;            // no level sums behind this pass
    scan_fill_list(ctx, 0, nb, fl);
    ctx->fs_ws.p = keep;
    launch_fill(fl, ctx->stream);
  }
  const uint32_t* wslot = nullptr;
  const uint32_t* rslot = nullptr;
  std::vector<uint32_t> spill;
  const int rc = scan_pass(ctx, 0, ctx->tnb.as<float>(), ctx->binmed.as<int32_t>(), nb, RDmedian, tmedian, tlamda, Lmax, ctx->status1.as<int32_t>(),
                           ctx->status1f.as<int32_t>(), &wslot, &rslot, &spill);
  if (rc != RSI_OK) return rc;
  HIPCHK(hipMemcpyAsync(status, ctx->status1.p, (size_t)nb * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(CTX_SYNC());
  if (info) { info[0] = (int32_t)wslot[8]; info[1] = (int32_t)wslot[0]; info[2] = (int32_t)wslot[1]; info[3] = 0; }
  return RSI_OK;
}

int rsi_hot_phase_times(const rsi_ctx* ctx, const char** names, double* ms, int cap) {
  if (!ctx) return 0;
  int k = 0;
  for (const auto& p : ctx->phases) { if (k < cap) { names[k] = p.first; ms[k] = p.second; } ++k; }
  return k;
}

int rsi_hot_kernel_times(const rsi_ctx* ctx, const char** names, float* ms, int cap) {
  if (!ctx) return 0;
  int k = 0;
  for (const KernelTime& t : ctx->ktimes) {
    if (k < cap) { names[k] = t.name; float v = 0; (void)hipEventElapsedTime(&v, t.a, t.b); ms[k] = v; }
    ++k;
  }
  return k;
}

// One output row as cnv_format1 prints it (rsi.cpp:581-631): default ostream formatting (%g-like,
// 6 significant digits); RP/Q0 are -1 on this path (rsi.h:49-50).
int rsi_result_format_row(const rsi_result* r, int i, const char* chrom, char* buf, int cap) {
  if (!r || i < 0 || i >= (int)r->lists[0].size()) return RSI_ERR_BAD_ARG;
  const rsi_call& c = r->lists[0][(size_t)i];
  static const char* kType[3] = {"DEL", "DUP", "UNKNOWN"};
  const int k = snprintf(buf, (size_t)cap, "%s\t%d\t%d\t%s\t%d\t%d\t%g(%g);%g(%g);%g(%g)\tRP=%d;Q0=%g\trsi", chrom, c.start, c.end,
                         kType[c.type < 0 || c.type > 2 ? 2 : c.type], c.qscore, c.end - c.start + 1, c.cnvmed,
                         c.cnviqr / 1.349, c.refmed, c.refiqr / 1.349, r->stats.RDmedian, r->stats.RDsd,
                         (size_t)i < r->rp.size() ? r->rp[(size_t)i] : -1, (size_t)i < r->q0.size() ? r->q0[(size_t)i] : -1.0);
  return k;
}

// The reference's per-L lines of rsistatus (rsi.cpp:1221-1224, 1251-1254), "DEL-\tL\tmarked so far\tbins\tportion", for the
// four sweeps in the order the reference runs them; line i into buf, returns i + 1, or 0 when there is no line i.
int rsi_result_log_line(const rsi_result* r, int i, char* buf, int cap) {
  // The (last) scan's diagnostic lines in the reference's order: the NB transform's two lines (rsi.cpp:1140-1141), the first
  // pass' DEL- / DUP+ lines per L (rsi.cpp:1221-1224, 1251-1254), filterstatus' level table (rsi.cpp:991-1002), the second
  // pass' per-L lines.
  if (!r || i < 0 || !buf || cap <= 0) return 0;
  int left = i;
  if (r->log_nb) {
    if (left == 0) { snprintf(buf, (size_t)cap, "RD median : %g", r->stats.RDmedian); return i + 1; }
    if (left == 1) { snprintf(buf, (size_t)cap, "RD median absolute deviation : %g", r->stats.nb_mad); return i + 1; }
    left -= 2;
  }
  for (int w = 0; w < 4; ++w) {
    if (w == 2) {
      if (left < (int)r->fs_lines.size()) { snprintf(buf, (size_t)cap, "%s", r->fs_lines[(size_t)left].c_str()); return i + 1; }
      left -= (int)r->fs_lines.size();
    }
    const int stop = (int)r->stop_levels[w];
    const std::vector<uint32_t>& h = r->level_log[w];
    if (stop <= 0 || h.empty()) continue;
    if (left >= stop) { left -= stop; continue; }
    const int L = left + 1;
    unsigned long long cum = 0;
    for (int l = 1; l <= L && l < (int)h.size(); ++l) cum += h[(size_t)l];
    const double portion = (double)(int)cum / (double)(int)r->stats.nbins;
    snprintf(buf, (size_t)cap, "%s\t%d\t%d\t%d\t%g", (w & 1) ? "DUP+" : "DEL-", L, (int)cum, (int)r->stats.nbins, portion);
    return i + 1;
  }
  return 0;
}

int rsi_result_summary(const rsi_result* r, int chrom_id, double* out, int max_calls) {
  if (!r || !out || max_calls < 0) return RSI_ERR_BAD_ARG;
  const std::vector<rsi_call>& L = r->lists[0];
  const int stored = (int)std::min<size_t>(L.size(), (size_t)max_calls);
  out[0] = (double)chrom_id; out[1] = r->stats.RDmedian; out[2] = r->stats.RDsd; out[3] = (double)L.size(); out[4] = (double)stored;
  out[5] = out[6] = out[7] = 0.0;
  for (int i = 0; i < stored; ++i) {
    const rsi_call& c = L[(size_t)i];
    double* o = out + RSI_SUMMARY_HEAD + RSI_SUMMARY_CALL * i;
    o[0] = c.start; o[1] = c.end; o[2] = c.type; o[3] = c.qscore; o[4] = c.cnvmed; o[5] = c.cnviqr; o[6] = c.refmed; o[7] = c.refiqr;
  }
  return RSI_SUMMARY_HEAD + RSI_SUMMARY_CALL * stored;
}

// One output row from a gathered summary block (what rank 0 of a multi-process run writes): same text as
// rsi_result_format_row on the rank that produced the block, RP / Q0 as on the depth-file path (-1).
int rsi_summary_format_row(const double* block, int i, const char* chrom, char* buf, int cap) {
  if (!block || !chrom || !buf || i < 0 || i >= (int)block[4]) return RSI_ERR_BAD_ARG;
  const double* o = block + RSI_SUMMARY_HEAD + RSI_SUMMARY_CALL * i;
  static const char* kType[3] = {"DEL", "DUP", "UNKNOWN"};
  const int type = (int)o[2];
  return snprintf(buf, (size_t)cap, "%s\t%d\t%d\t%s\t%d\t%d\t%g(%g);%g(%g);%g(%g)\tRP=%d;Q0=%g\trsi", chrom, (int)o[0], (int)o[1],
                  kType[type < 0 || type > 2 ? 2 : type], (int)o[3], (int)o[1] - (int)o[0] + 1, o[4], o[5] / 1.349, o[6], o[7] / 1.349,
                  block[1], block[2], -1, -1.0);
}

// All rows of a block at once, each ended by a newline; returns the number of bytes written (0: no calls), < 0 on error
// (also when buf is too small).
int rsi_summary_format_rows(const double* block, const char* chrom, char* buf, int cap) {
  if (!block || !chrom || !buf || cap <= 0) return RSI_ERR_BAD_ARG;
  const int stored = (int)block[4];
  int used = 0;
  for (int i = 0; i < stored; ++i) {
    const int k = rsi_summary_format_row(block, i, chrom, buf + used, cap - used - 1);
    if (k < 0 || k >= cap - used - 1) return RSI_ERR_BAD_ARG;
    used += k;
    buf[used++] = '\n';
  }
  buf[used < cap ? used : cap - 1] = 0;
  return used;
}

int rsi_result_pairs(const rsi_result* r, int i, int32_t* rp, double* q0) {
  if (!r || i < 0 || i >= (int)r->lists[0].size()) return RSI_ERR_BAD_ARG;
  if (rp) *rp = (size_t)i < r->rp.size() ? r->rp[(size_t)i] : -1;
  if (q0) *q0 = (size_t)i < r->q0.size() ? r->q0[(size_t)i] : -1.0;
  return RSI_OK;
}

int rsi_result_annotate_bam(rsi_result* r, const char* bam_path, const char* chrom) {
  if (!r || !bam_path || !chrom) return RSI_ERR_BAD_ARG;
  std::string err;
  rsih::BamFile bam;
  std::vector<std::pair<std::string, int64_t>> refs;
  uint64_t voff = 0;
  if (!bam.open(bam_path, err) || !bam.read_header(refs, voff, err)) { set_global_error(err); return RSI_ERR_BAD_ARG; }
  int tid = -1;
  for (size_t k = 0; k < refs.size(); ++k) if (refs[k].first == chrom) tid = (int)k;
  if (tid < 0) { set_global_error(std::string("chromosome not in the BAM header: ") + chrom); return RSI_ERR_BAD_ARG; }
  const std::string bai = std::string(bam_path) + ".bai";
  rsih::PairSample ps;
  if (!rsih::bam_pair_sample(bam, bai, tid, refs[(size_t)tid].second, 10000000, 349250621, ps, err)) { set_global_error(err); return RSI_ERR_BAD_ARG; }
  std::vector<rsih::CallSpan> spans;
  for (const rsi_call& c : r->lists[0]) spans.push_back({c.start, c.end, c.type, -1, -1.0});
  if (!rsih::bam_annotate_calls(bam, bai, tid, ps, spans, err)) { set_global_error(err); return RSI_ERR_BAD_ARG; }
  r->rp.clear(); r->q0.clear();
  for (const rsih::CallSpan& c : spans) { r->rp.push_back(c.rp); r->q0.push_back(c.q0); }
  return RSI_OK;
}

}  // extern "C"
