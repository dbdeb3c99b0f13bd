// host_calls.h -- host-sized stages of the path: candidate tests,
// boundary refinement, merge, final filters.  Candidates are few (10^2..10^4) and the walks are
// sequential and data-dependent, so these stay on the host; the per-base depth they read stays in
// HBM and is paged in on demand (DepthPager).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <memory>
#include <vector>
#include "../../include/rsi_hot.h"

namespace rsih {

enum { kDel = 0, kDup = 1, kUnknown = 2 };   // rsi.h:4-6

struct Candidate {     // working record, mirrors cnv_st (rsi.h:8-51)
  int type = kUnknown, geno = 0, status = 0, start = 0, end = 0, length = 0;
  double score = 0, p1 = 1.0, cnvmed = 0, cnvsd = 0, cnviqr = 0, refmed = 0, refsd = 0, refiqr = 0;
};

struct Region { int start, end; };   // inclusive, reference coordinates (rsi::noncodelist)

// Read-only view of an int32 array living in device memory, fetched in 64 KB pages on first touch.
class DepthPager {
 public:
  DepthPager(const int32_t* d_ptr, int64_t n, hipStream_t stream);
  int64_t size() const { return n_; }
  int operator[](int64_t i) { return page(i >> kBits)[i & kMask]; }
  void prefetch(int64_t lo, int64_t hi);   // [lo, hi] clipped to the array, one copy per missing stretch
  int64_t bytes_fetched() const { return fetched_; }
 private:
  static constexpr int kBits = 14;
  static constexpr int64_t kMask = (1 << kBits) - 1;
  const int32_t* page(int64_t p) { if (!pages_[p]) fetch(p, p); return pages_[p].get(); }
  void fetch(int64_t p0, int64_t p1);
  const int32_t* d_;
  int64_t n_;
  hipStream_t stream_;
  std::vector<std::unique_ptr<int32_t[]>> pages_;
  int64_t fetched_ = 0;
};

struct CallerInput {
  rsi_params P;
  double RDmedian, RDsd;
  int64_t ncompact;                 // rsi::end with rsi::start = 1
  const std::vector<Region>* noncode;
  const std::vector<int>* binmedint;
};

// areblockscnv on one scan's segments (rsi.cpp:415-546); segs are updated in place.
void test_block_segments(const CallerInput& in, const std::vector<int>& status, std::vector<Candidate>& segs);

// Everything detectcnv does after the block tests (rsi.cpp:1860-1931) plus sd_filters
// (rsi.cpp:1753-1792).  `blocks` receives the sorted bin-space list, `raw` the calls before
// sd_filters, `kept` the final calls.
void call_from_segments(const CallerInput& in, std::vector<Candidate> segs, DepthPager& depth,
                        std::vector<Candidate>& blocks, std::vector<Candidate>& raw, std::vector<Candidate>& kept);

}  // namespace rsih
