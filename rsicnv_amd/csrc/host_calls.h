// host_calls.h -- the candidate stages of the path: candidate tests,
// boundary refinement, merge, final filters.  Candidates are few (10^2..10^4) and the list logic
// around them is sequential and data-dependent: that part lives here, on the host.  Everything
// that touches the per-base depth runs on the device through NeighbourTester (kernels_cand.hip,
// implemented next to the pipeline); the host walks below remain as the path a test takes when
// the device declines it, reading the depth through DepthPager (pages fetched from HBM on demand).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <memory>
#include <utility>
#include <vector>
#include "../../include/rsi_hot.h"

namespace rsih {

enum { kDel = 0, kDup = 1, kUnknown = 2 };   // rsi.h:4-6

struct Candidate {     // working record, mirrors cnv_st (rsi.h:8-51)
  int type = kUnknown, geno = 0, status = 0, start = 0, end = 0, length = 0;
  double score = 0, p1 = 1.0, cnvmed = 0, cnvsd = 0, cnviqr = 0, refmed = 0, refsd = 0, refiqr = 0;
};

struct Region { int start, end; };   // inclusive, reference coordinates (rsi::noncodelist)

// Read-only view of an int32 array living in device memory.  A host mirror of the whole array is
// reserved but never touched up front; 64 KB pages are copied from HBM into it on first use
// (about 0.1-1 % of a chromosome around the candidates), in one transfer per missing stretch.
class DepthPager {
 public:
  // mirror: provider of a host buffer of at least n elements owned by the caller (reused across
  // chromosomes, so no address space is mapped and unmapped per call), asked for on the first page
  // fetch only -- with the candidate stages on the device it is normally never needed; staging:
  // optional pinned buffer for the DMA.
  DepthPager(const int32_t* d_ptr, int64_t n, hipStream_t stream, std::function<int32_t*()> mirror, void* staging = nullptr,
             size_t staging_bytes = 0, hipEvent_t sync_event = nullptr);
  // The array already sits in host memory (the CPU harness of the candidate stages, tests/sanitize): no pages, no copies.
  DepthPager(const int32_t* host_ptr, int64_t n);
  DepthPager(const DepthPager&) = delete;
  DepthPager& operator=(const DepthPager&) = delete;
  int64_t size() const { return n_; }
  int operator[](int64_t i) {
    if (!have_[(size_t)(i >> kBits)]) fetch(i >> kBits, i >> kBits);
    return failed_ ? 0 : mirror_[i];
  }
  bool failed() const { return failed_; }   // no host memory for the mirror: every value read since was 0, the results are void
  void prefetch(int64_t lo, int64_t hi);   // [lo, hi] clipped to the array, one copy per missing stretch
  const int32_t* raw() const { return mirror_; }   // valid only inside prefetched ranges
  int64_t bytes_fetched() const { return fetched_; }
  double fetch_ms() const { return fetch_ms_; }
 private:
  static constexpr int kBits = 14;
  void fetch(int64_t p0, int64_t p1);
  const int32_t* d_;
  int64_t n_;
  hipStream_t stream_;
  std::function<int32_t*()> mirror_source_;
  int32_t* mirror_ = nullptr;          // obtained from mirror_source_ on the first fetch
  std::vector<unsigned char> have_;
  int32_t* staging_ = nullptr;         // pinned, owned by the context
  int64_t staging_elems_ = 0;
  hipEvent_t sync_ev_ = nullptr;       // blocking-sync event for the waits (no busy spinning), optional
  void wait();
  int64_t fetched_ = 0;
  double fetch_ms_ = 0;
  bool failed_ = false;
};

// wall-clock split of the candidate stages (ms), filled when CallerInput::prof is set
struct CallProfile {
  double fetch = 0, gather = 0, winmean = 0, quantiles = 0, variance = 0, sharpen = 0, merge = 0, final_tests = 0, device_ms = 0;
  int tests = 0, spec_hits = 0, single_tests = 0, host_fallbacks = 0, block_batch_hits = 0;
};

// Statistics of one neighbourhood test (what isitcnv derives from its two arrays, rsi.cpp:113-147).
struct TestStats { double cnv_lqt, cnv_med, cnv_uqt, cnv_var, ref_lqt, ref_med, ref_uqt, ref_var; };

// One prepared neighbourhood test: the part of isitcnvwrap (rsi.cpp:175-257) that only looks at
// the candidate list -- sizes, margins and the neighbour intervals the two walks may jump over, in
// the order the reference's `idx` pointer visits them.
struct TestPlan {
  int start, end, kind, margin, capacity, top, budget, cut;
  double right_cap;
  std::vector<std::pair<int, int>> left_chain, right_chain;   // (start, end)
};

// Per-base tests executed on the device (kernels_cand.hip); implemented next to the pipeline.
class NeighbourTester {
 public:
  virtual ~NeighbourTester() {}
  // optimize_with_derivative twice for every candidate (rsi.cpp:1876-1877)
  virtual bool sharpen(std::vector<Candidate>& L) = 0;
  // ok[k] = 0: the device could not serve plan k (histogram range); the caller uses the host path
  virtual bool test(const std::vector<TestPlan>& plans, std::vector<TestStats>& stats, std::vector<int>& left_reach,
                    std::vector<char>& ok) = 0;
  // exact sums of the depth over inclusive ranges (mean_tp in mergesegments, rsi.cpp:775-779)
  virtual bool range_sums(const std::vector<std::pair<int, int>>& ranges, std::vector<int64_t>& sums) = 0;
};

// A bin-space int array the caller keeps alive (the pinned host copy of a device array, or a vector): read in place.
struct IntSpan {
  const int* p = nullptr;
  int64_t n = 0;
  // sparse form: only the values inside some index ranges are present (the status array inside the marked runs, which is
  // all the host reads of it): ranges[k] = (first index, last index, position of the first value in p), sorted by index
  struct Range { int64_t first, last, at; };
  const std::vector<Range>* ranges = nullptr;
  IntSpan() {}
  IntSpan(const int* p_, int64_t n_) : p(p_), n(n_) {}
  IntSpan(const std::vector<int>& v) : p(v.data()), n((int64_t)v.size()) {}
  IntSpan(const int* values, int64_t n_, const std::vector<Range>* r) : p(values), n(n_), ranges(r) {}
  const int* at(int64_t i) const {   // the value of index i (and of the indices behind it up to the end of its range)
    if (!ranges) return p + i;
    size_t lo = 0, hi = ranges->size();
    while (lo + 1 < hi) { const size_t mid = (lo + hi) / 2; if ((*ranges)[mid].first <= i) lo = mid; else hi = mid; }
    const Range& r = (*ranges)[lo];
    if (i < r.first || i > r.last) { static const int zero = 0; return &zero; }   // outside the runs the status is 0
    return p + r.at + (i - r.first);
  }
  int operator[](int64_t i) const { return ranges ? *at(i) : p[i]; }
  int64_t size() const { return n; }
};

struct CallerInput {
  rsi_params P;
  CallProfile* prof = nullptr;
  NeighbourTester* tester = nullptr;   // when set, per-base candidate work runs on the device
  NeighbourTester* block_tester = nullptr;   // the same on the BIN medians (areblockscnv): one batch for a scan's segments when there are many
  double RDmedian, RDsd;
  int64_t ncompact;                 // rsi::end with rsi::start = 1
  const std::vector<Region>* noncode;
  IntSpan binmedint;
  // Tests whose neighbourhood is no longer than the candidate itself (the walks ran out of sequence around a candidate that
  // spans most of what is left): the reference sizes its running-mean array ref - body (rsi.cpp:107) and indexes it -- its
  // Array throws and the program aborts.  Counted here; the caller fails the chromosome with RSI_ERR_UNSUPPORTED.
  int* short_neighbourhoods = nullptr;
};

// areblockscnv on one scan's segments (rsi.cpp:415-546); segs are updated in place.
void test_block_segments(const CallerInput& in, IntSpan status, std::vector<Candidate>& segs);

// Everything detectcnv does after the block tests (rsi.cpp:1860-1931) plus sd_filters
// (rsi.cpp:1753-1792).  `blocks` receives the sorted bin-space list, `raw` the calls before
// sd_filters, `kept` the final calls.
void call_from_segments(const CallerInput& in, std::vector<Candidate> segs, DepthPager& depth,
                        std::vector<Candidate>& blocks, std::vector<Candidate>& raw, std::vector<Candidate>& kept);

// Depth from host memory, narrowed to bytes before it crosses PCIe: dst[i] = src[i] where 0 <= src[i] < 255, else 255 with the value
// listed in (esc_pos, esc_val) (up to cap entries).  Returns the number of values that did not fit (may exceed cap).
int64_t narrow_depth_u8(const int32_t* src, int64_t n, uint8_t* dst, int32_t* esc_pos, int32_t* esc_val, int64_t cap);

}  // namespace rsih
