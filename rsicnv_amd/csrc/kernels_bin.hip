// kernels_bin.hip -- bin-level kernels (gfx950): NB transform, 0.01-grid quantile histograms,
// the RSI scan with its LDS prefix tile, status resolution, marked runs and the max-score
// sub-segment search.  These work on nb = n'/m bins (about 1 % of the bases), so they are
// latency/ALU-bound rather than HBM-bound; DESIGN.md section 4 gives each one's budget.
//
// Built with -ffp-contract=off and IEEE division/sqrt: every double expression below has to round
// like the reference's x86-64 build.
#include <type_traits>
#include "kernels.h"
#include "device_util.h"

namespace rsik {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxGrid = 256 * 2;   // grid-stride kernels; most end with one same-address arrival atomic per workgroup (12 ns each)

__device__ inline int lane_id() { return threadIdx.x & 63; }
inline int grid_for(int64_t items, int per_block) {
  int64_t g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

// total order on float bit patterns (so that min/max can use integer atomics)
__device__ inline uint32_t f32_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// ------------------------------------------------------------------------------------------
// K5  NB transform (rsi.cpp:1155-1162): the formula, then the running minimum.
__global__ __launch_bounds__(kThreads) void k_nb_raw(const int64_t* __restrict__ binsum, int64_t nb, int m,
                                                     int64_t ncompact, double r, float* __restrict__ raw,
                                                     uint32_t* __restrict__ rawmin_key) {
  uint32_t kmin = 0xffffffffu;
  for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < nb; b += (int64_t)gridDim.x * kThreads) {
    const int64_t i1 = b * m;
    int64_t i2 = b * m + m - 1;
    if (i2 > ncompact - 1) i2 = ncompact - 1;
    const double m2 = (double)(i2 - i1 + 1);
    const double sum = (double)binsum[b];
    const double q = (sum + 0.25) / (m2 * r - 0.5);
    const double t = 2.0 * sqrt(r) * log(sqrt(q) + sqrt(1.0 + q));
    const float f = (float)t;
    raw[b] = f;
    const uint32_t k = f32_key(f);
    kmin = k < kmin ? k : kmin;
  }
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(kmin, d); kmin = o < kmin ? o : kmin; }
  // one atomic per workgroup of a bounded grid: one per wave of one-bin threads was 19 000 atomics on the same word per
  // chromosome, and those serialise (the kernel took 100 us for 1.2 M bins, 80 of them in that queue)
  __shared__ uint32_t s_min[kThreads / 64];
  if (lane_id() == 0) s_min[threadIdx.x >> 6] = kmin;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) kmin = s_min[w] < kmin ? s_min[w] : kmin;
    if (kmin != 0xffffffffu) atomicMax(rawmin_key, ~kmin);   // complement: all zero = nothing seen
  }
}

// ------------------------------------------------------------------------------------------
// K6  quantile histograms of float arrays on the 0.01 grid (partition_stat_tp, wufunctions.cpp:364-424)
__device__ inline float sel_value(const float* __restrict__ x, int64_t i, int use_abs, double center) {
  const float v = x[i];
  return use_abs ? (float)fabs((double)v - center) : v;   // RDtmp[i] = abs(RDtrans[i]-tmedian), rsi.cpp:1276
}

__device__ __forceinline__ void minmax_body(const float* __restrict__ x, const int32_t* __restrict__ mask, int64_t nb,
                                            int use_abs, double center, MinMaxF* __restrict__ mm) {
  uint32_t kmin = 0xffffffffu, kmax = 0;
  unsigned int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nb; i += (int64_t)gridDim.x * kThreads) {
    if (mask && mask[i] != 0) continue;
    const float v = sel_value(x, i, use_abs, center);
    if (!(fabsf(v) <= 3.0e38f)) bad = 1;
    const uint32_t k = f32_key(v);
    kmin = k < kmin ? k : kmin;
    kmax = k > kmax ? k : kmax;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t a = __shfl_xor(kmin, d), b = __shfl_xor(kmax, d);
    kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; bad |= __shfl_xor(bad, d);
  }
  __shared__ uint32_t s_min[kThreads / 64], s_max[kThreads / 64], s_bad[kThreads / 64];
  if (lane_id() == 0) { s_min[threadIdx.x >> 6] = kmin; s_max[threadIdx.x >> 6] = kmax; s_bad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {   // one set of atomics per workgroup
    for (int w = 1; w < kThreads / 64; ++w) { kmin = s_min[w] < kmin ? s_min[w] : kmin; kmax = s_max[w] > kmax ? s_max[w] : kmax; bad |= s_bad[w]; }
    if (kmin != 0xffffffffu) atomicMax(&mm->min_inv, ~kmin);
    if (kmax != 0) atomicMax(&mm->max_bits, kmax);
    if (bad) atomicOr(&mm->nonfinite, 1u);
  }
}

__global__ __launch_bounds__(kThreads) void k_minmax_f32(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                         int64_t nb, int use_abs, double center, MinMaxF* __restrict__ mm) {
  minmax_body(x, mask, nb, use_abs, center, mm);
}
constexpr uint32_t kLdsBins = 12288;   // 48 KB of LDS counters: covers a value range of 122 from the minimum; the buckets behind go
                                       // to global atomics (few values live there).  With 128 KB the kernel only fitted on a CU no per-base
                                       // kernel of another chromosome was resident on, and waited for one: covers a value range of 327

// Each thread takes kHistRun consecutive elements and merges equal neighbouring buckets before the
// atomic: bin medians (-MED) are small integers, so long runs land in one bucket and a plain
// one-atomic-per-element histogram serialises on it (271 ms per 3 Gb step with global atomics).
constexpr int kHistRun = 8;

// PACK16: two 16-bit LDS counters to a word (half the LDS: the kernel then fits beside four resident per-base workgroups
// instead of waiting for one to leave); the caller guarantees that a workgroup counts fewer than 65536 values in all.
template <bool PACK16 = false>
__device__ __forceinline__ void hist_body(const float* __restrict__ x, const int32_t* __restrict__ mask, int64_t nb, int use_abs,
                                          double center, double ymin, uint32_t* __restrict__ hist, uint32_t np, int use_lds,
                                          unsigned int* s_h) {
  const uint32_t nl = use_lds ? (np < (uint32_t)use_lds ? np : (uint32_t)use_lds) : 0;   // buckets counted in LDS (use_lds = how many fit)
  auto lds_add = [&](uint32_t k, uint32_t c) { if (PACK16) atomicAdd(&s_h[k >> 1], c << ((k & 1u) << 4)); else atomicAdd(&s_h[k], c); };
  auto bucket = [&](float v) {
    const double idx = ((double)v - ymin) / 0.01 + 0.5;        // wufunctions.cpp:396
    uint32_t k = (uint32_t)(unsigned long long)idx;
    return k >= np ? np - 1 : k;                               // cannot happen (np = range/dy + 2)
  };
  // Which nl buckets live in LDS: [wb, wb + nl).  A grid longer than that (values spread over more than 122: deep coverage --
  // at 300x every bin's value lay beyond the first 12 288 buckets and took a global atomic, 280 us per launch instead of 40)
  // gets its window around the median of 64 values sampled across the array; every workgroup samples the same positions.
  __shared__ uint32_t s_wb;
  if (threadIdx.x == 0) s_wb = 0u;
  if (use_lds) { for (uint32_t e = threadIdx.x; e < (PACK16 ? (nl + 1) / 2 : nl); e += kThreads) s_h[e] = 0; }
  if (use_lds && np > nl && threadIdx.x < 64) {
    const int lane = (int)threadIdx.x;
    const int64_t i = (int64_t)(((unsigned long long)(2 * lane + 1) * (unsigned long long)nb) >> 7);   // the middle of the lane's 64th of the array
    const bool valid = i < nb && !(mask && mask[i] != 0);
    const uint32_t k = valid ? bucket(sel_value(x, i, use_abs, center)) : 0xffffffffu;
    int rank = 0;
    for (int j = 0; j < 64; ++j) {
      const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)k, j);
      rank += (kj < k || (kj == k && j < lane)) ? 1 : 0;     // invalid samples (all ones) rank last
    }
    const int nvalid = __popcll(__ballot(valid));
    if (valid && rank == nvalid / 2) s_wb = k < nl / 2 ? 0u : (k - nl / 2 > np - nl ? np - nl : k - nl / 2);
  }
  if (use_lds) __syncthreads();
  const uint32_t wb = use_lds ? s_wb : 0u;
  auto count = [&](uint32_t k, uint32_t c) { if (k - wb < nl) lds_add(k - wb, c); else atomicAdd(&hist[k], c); };
  const int64_t nchunks = (nb + kHistRun - 1) / kHistRun;
  for (int64_t c = (int64_t)blockIdx.x * kThreads + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * kThreads) {
    uint32_t pend_k = 0xffffffffu, pend_c = 0;
    const int64_t i0 = c * kHistRun, i1 = i0 + kHistRun < nb ? i0 + kHistRun : nb;
    auto one = [&](float xv) {
      const float v = use_abs ? (float)fabs((double)xv - center) : xv;   // sel_value
      const uint32_t k = bucket(v);
      if (k != pend_k) {
        if (pend_c) count(pend_k, pend_c);
        pend_k = k; pend_c = 0;
      }
      ++pend_c;
    };
    if (i1 - i0 == kHistRun) {
      // a whole run: its eight values (and mask words) as 16-byte loads, all in flight at once -- element by element the loop
      // paid a memory latency per value, most of the kernel's 37 us on a 60 Mb chromosome
      static_assert(kHistRun == 8, "two quads per run");
      const float4 a = *reinterpret_cast<const float4*>(x + i0), b = *reinterpret_cast<const float4*>(x + i0 + 4);
      int4 ma = make_int4(0, 0, 0, 0), mb = ma;
      if (mask) { ma = *reinterpret_cast<const int4*>(mask + i0); mb = *reinterpret_cast<const int4*>(mask + i0 + 4); }
      const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const int mv[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) if (mv[e] == 0) one(xv[e]);
    } else {
      for (int64_t i = i0; i < i1; ++i) {
        if (mask && mask[i] != 0) continue;
        one(x[i]);
      }
    }
    if (pend_c) count(pend_k, pend_c);
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nl; e += kThreads) {
      const unsigned int c = PACK16 ? (s_h[e >> 1] >> ((e & 1u) << 4)) & 0xffffu : s_h[e];
      if (c) atomicAdd(&hist[wb + e], c);
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_hist_f32(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                       int64_t nb, int use_abs, double center, double ymin,
                                                       uint32_t* __restrict__ hist, uint32_t np, int use_lds) {
  extern __shared__ unsigned int s_h[];
  hist_body(x, mask, nb, use_abs, center, ymin, hist, np, use_lds, s_h);
}

// ---- the same median as a device-side chain of TWO launches: (min/max, and the last workgroup to finish derives the
// grid and clears its buckets) -> (histogram, and the last workgroup walks it to the median).  Nothing returns to the
// host in between; a second chain takes its centre from the first one's result (median, then MAD).  The tests and
// their order are grid_median()'s (pipeline.hip). ----
__device__ inline float f32_unkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// one workgroup, after every min/max atomic of the launch has landed: anchor and bucket count -> *g, buckets cleared,
// the min/max record back to "nothing seen"
__device__ inline void grid_plan_block(MinMaxF* __restrict__ mm, uint32_t cap, uint32_t* __restrict__ hist, GridMedian* __restrict__ g) {
  const uint32_t kmin = ~ld_cg(&mm->min_inv), kmax = ld_cg(&mm->max_bits), bad = ld_cg(&mm->nonfinite);
  uint32_t flags = 0, np = 0;
  double ymin = 0.0;
  if (kmin == 0xffffffffu) flags = kGridEmpty;
  else if (bad) flags = kGridNonFinite;
  else {
    ymin = (double)f32_unkey(kmin);
    const double ymax = (double)f32_unkey(kmax);
    if ((ymax - ymin) < 0.01) flags = kGridDegenerate;
    else {
      const size_t n = (size_t)((ymax - ymin) / 0.01 + 2);
      if (n > cap) flags = kGridTooWide; else np = (uint32_t)n;
    }
  }
  for (uint32_t e = threadIdx.x; e < np; e += kThreads) hist[e] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    g->med = 0.0; g->ymin = ymin; g->count = 0; g->np = np; g->flags = flags;
    mm->min_inv = 0; mm->max_bits = 0; mm->nonfinite = 0;
  }
}

// min/max of a workgroup's values -> the record (one set of atomics per workgroup)
__device__ inline void minmax_commit(uint32_t kmin, uint32_t kmax, unsigned int bad, MinMaxF* __restrict__ mm) {
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t a = __shfl_xor(kmin, d), b = __shfl_xor(kmax, d);
    kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax; bad |= __shfl_xor(bad, d);
  }
  __shared__ uint32_t s_min[kThreads / 64], s_max[kThreads / 64], s_bad[kThreads / 64];
  if (lane_id() == 0) { s_min[threadIdx.x >> 6] = kmin; s_max[threadIdx.x >> 6] = kmax; s_bad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) { kmin = s_min[w] < kmin ? s_min[w] : kmin; kmax = s_max[w] > kmax ? s_max[w] : kmax; bad |= s_bad[w]; }
    if (kmin != 0xffffffffu) atomicMax(&mm->min_inv, ~kmin);
    if (kmax != 0) atomicMax(&mm->max_bits, kmax);
    if (bad) atomicOr(&mm->nonfinite, 1u);
  }
}

__global__ __launch_bounds__(kThreads) void k_minmax_plan(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                          int64_t nb, int use_abs, double center, const double* __restrict__ center_ptr,
                                                          MinMaxF* __restrict__ mm, unsigned int* __restrict__ counter, uint32_t cap,
                                                          uint32_t* __restrict__ hist, GridMedian* __restrict__ g) {
  const double c = center_ptr ? *center_ptr : center;
  uint32_t kmin = 0xffffffffu, kmax = 0;
  unsigned int bad = 0;
  auto one = [&](float xv) {
    const float v = use_abs ? (float)fabs((double)xv - c) : xv;   // sel_value
    if (!(fabsf(v) <= 3.0e38f)) bad = 1;
    const uint32_t k = f32_key(v);
    kmin = k < kmin ? k : kmin;
    kmax = k > kmax ? k : kmax;
  };
  // four of a thread's strided elements per trip, their loads in flight together (one load and one wait per element was most of
  // this kernel's 13 us)
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  for (; i + 3 * stride < nb; i += 4 * stride) {
    float xv[4]; int mv[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] = x[i + e * stride];
    if (mask) {
#pragma unroll
      for (int e = 0; e < 4; ++e) mv[e] = mask[i + e * stride];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) if (mv[e] == 0) one(xv[e]);
  }
  for (; i < nb; i += stride) {
    if (mask && mask[i] != 0) continue;
    one(x[i]);
  }
  minmax_commit(kmin, kmax, bad, mm);
  if (last_block_done(counter)) grid_plan_block(mm, cap, hist, g);
}

// What the host picks up after a chain: results and scalars that sit in one block of device memory, copied to mapped host
// memory by the chain's last workgroup.
struct ExportPair { const void* src[2]; void* dst[2]; unsigned int bytes[2]; };

// hist_median_grid (hostmath.h) by one workgroup: the bucket in which the running count first reaches total/2.
// Two sweeps of coherent loads instead of one dependent load per bucket: every thread sums a contiguous stretch of buckets with
// 16-byte loads, eight in flight; the stretch in which the count crosses total/2 is then read once more by the whole workgroup,
// a few buckets per thread.  (One load and one wait per bucket was 30 of this kernel's 37 us on a 30x chromosome -- 12 000
// buckets of 0.01 -- and 300 us at 300x, where the values span ten times the range.)
__device__ inline void grid_walk_block(const uint32_t* __restrict__ hist, GridMedian* __restrict__ g) {
  __shared__ unsigned long long s_sum[kThreads];
  __shared__ unsigned long long s_seen;
  __shared__ unsigned int s_cross;
  const uint32_t np = g->np;
  const uint32_t chunk = (((np + kThreads - 1) / kThreads) + 3u) & ~3u;   // whole quads
  const uint32_t b0 = threadIdx.x * chunk < np ? threadIdx.x * chunk : np;
  const uint32_t b1 = b0 + chunk < np ? b0 + chunk : np;
  const __amdgpu_buffer_rsrc_t rs = coherent_buffer(hist);
  unsigned long long mine = 0;
  for (uint32_t b = b0; b < b1; b += 32) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint32_t q = b + 4u * j; v[j] = ld_cg_buf_x4(rs, (q < b1 ? q : b0) * 4u, 0u); }   // (past the stretch: a valid quad, not counted)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t q = b + 4u * j;
      const unsigned int c[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) mine += q + e < b1 ? c[e] : 0u;
    }
  }
  s_sum[threadIdx.x] = mine;
  if (threadIdx.x == 0) s_cross = 0xffffffffu;
  __syncthreads();
  unsigned long long seen = 0, total = 0;
  for (int t = 0; t < kThreads; ++t) { const unsigned long long v = s_sum[t]; if (t < (int)threadIdx.x) seen += v; total += v; }
  const unsigned long long r2 = total / 2;
  // agent-scope stores: the same workgroup reads the record back through ld_cg for the export
  unsigned long long* med_bits = reinterpret_cast<unsigned long long*>(&g->med);
  if (threadIdx.x == 0) { st_cg(&g->count, total); if (r2 == 0) st_cg(med_bits, (unsigned long long)__double_as_longlong(g->ymin)); }
  if (r2 != 0 && mine != 0 && seen < r2 && seen + mine >= r2) { s_cross = b0; s_seen = seen; }   // exactly one thread
  __syncthreads();
  const uint32_t cb = s_cross;
  if (cb == 0xffffffffu) return;   // (uniform)
  const unsigned long long before = s_seen;
  const uint32_t ce = cb + chunk < np ? cb + chunk : np;
  const uint32_t per = (chunk + kThreads - 1) / kThreads;   // buckets of the crossing stretch per thread: 1 .. 16
  const uint32_t m0 = cb + threadIdx.x * per < ce ? cb + threadIdx.x * per : ce;
  const uint32_t m1 = m0 + per < ce ? m0 + per : ce;
  unsigned long long part = 0;
  for (uint32_t bb = m0; bb < m1; ++bb) part += ld_cg(hist + bb);
  __syncthreads();   // s_sum is read no more
  s_sum[threadIdx.x] = part;
  __syncthreads();
  unsigned long long run = before;
  for (int t = 0; t < (int)threadIdx.x; ++t) run += s_sum[t];
  if (part != 0 && run < r2 && run + part >= r2) {
    for (uint32_t bb = m0; bb < m1; ++bb) {
      const unsigned long long upto = run + ld_cg(hist + bb);
      if (run < r2 && upto >= r2) st_cg(med_bits, (unsigned long long)__double_as_longlong(g->ymin + (double)bb * 0.01));
      run = upto;
    }
  }
}

template <bool PACK16>
__global__ __launch_bounds__(kThreads) void k_hist_walk(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                        int64_t nb, int use_abs, double center,
                                                        const double* __restrict__ center_ptr,
                                                        GridMedian* __restrict__ g, uint32_t* __restrict__ hist,
                                                        unsigned int* __restrict__ counter, ExportPair ex, FillList fill, uint32_t lds_bins) {
  extern __shared__ unsigned int s_h[];
  fill_ranges(fill);   // for the kernels behind this one (the scan's first-L arrays, its counters)
  if (!g->flags) {
    const uint32_t np = g->np;
    hist_body<PACK16>(x, mask, nb, use_abs, center_ptr ? *center_ptr : center, g->ymin, hist, np, (int)lds_bins, s_h);   // the first lds_bins buckets in LDS
  }
  if (!last_block_done(counter)) return;
  if (!g->flags) grid_walk_block(hist, g);
  sync_drained();
  for (int k = 0; k < 2; ++k) export_words(ex.dst[k], ex.src[k], ex.bytes[k]);
}

// rsi.cpp:1176-1185: subtract the minimum, rescale to the depth scale, overwrite bins 0..2 -- and, while the values
// pass through, their min/max for the median that follows (the last workgroup derives that median's grid).
// The scalars come from the raw minimum found by k_nb_raw and the three reference levels the host computed with its
// own libm (no device log in them); the host repeats the same IEEE operations on the minimum it gets back.
struct NbLevels { double med_raw, del_raw, dup_raw, RDmedian; };
__global__ __launch_bounds__(kThreads) void k_nb_scale_mm(float* __restrict__ x, int64_t nb, const uint32_t* __restrict__ rawmin_inv,
                                                          NbLevels lv, MinMaxF* __restrict__ mm, unsigned int* __restrict__ counter,
                                                          uint32_t cap, uint32_t* __restrict__ hist, GridMedian* __restrict__ g) {
  const uint32_t key = ~(*rawmin_inv);
  const double tmin = (double)__uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
  const double med_nbt = lv.med_raw - tmin;
  const double del_s = (lv.del_raw - tmin) / med_nbt * lv.RDmedian, dup_s = (lv.dup_raw - tmin) / med_nbt * lv.RDmedian;
  const double med_s = med_nbt / med_nbt * lv.RDmedian;
  const float lev0 = (float)del_s, lev1 = (float)dup_s, lev2 = (float)med_s;
  uint32_t kmin = 0xffffffffu, kmax = 0;
  unsigned int bad = 0;
  for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < nb; b += (int64_t)gridDim.x * kThreads) {
    float v = x[b];
    v = (float)((double)v - tmin);
    v = (float)((double)v / med_nbt * lv.RDmedian);
    if (b == 0) v = lev0;
    if (b == 1) v = lev1;
    if (b == 2) v = lev2;
    x[b] = v;
    if (!(fabsf(v) <= 3.0e38f)) bad = 1;
    const uint32_t k = f32_key(v);
    kmin = k < kmin ? k : kmin;
    kmax = k > kmax ? k : kmax;
  }
  minmax_commit(kmin, kmax, bad, mm);
  if (last_block_done(counter)) grid_plan_block(mm, cap, hist, g);
}

// -MED: the bin medians as floats, and the min/max of their absolute deviations from `center` for the MAD that follows
__global__ __launch_bounds__(kThreads) void k_i32_to_f32_mm(const int32_t* __restrict__ in, float* __restrict__ out, int64_t nb,
                                                            double center, MinMaxF* __restrict__ mm, unsigned int* __restrict__ counter,
                                                            uint32_t cap, uint32_t* __restrict__ hist, GridMedian* __restrict__ g) {
  uint32_t kmin = 0xffffffffu, kmax = 0;
  unsigned int bad = 0;
  for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < nb; b += (int64_t)gridDim.x * kThreads) {
    const float f = (float)in[b];
    out[b] = f;
    const float v = (float)fabs((double)f - center);
    if (!(fabsf(v) <= 3.0e38f)) bad = 1;
    const uint32_t k = f32_key(v);
    kmin = k < kmin ? k : kmin;
    kmax = k > kmax ? k : kmax;
  }
  minmax_commit(kmin, kmax, bad, mm);
  if (last_block_done(counter)) grid_plan_block(mm, cap, hist, g);
}

// ------------------------------------------------------------------------------------------
// K7  RSI scan.  One workgroup per tile of kScanTile bins (one per thread).  The tile plus a halo of
// Lmax/2+1 bins each side is staged in LDS as prefix arrays only:
//   P    exact double prefix of the transformed values: the window sum of any (bin, L) is one subtraction;
//   CL/CG   counts of bins with median <= floor(0.75 RDmedian) / >= ceil(1.25 RDmedian): the exact
//        window-median test is a count difference (the two middle elements are needed only when the
//        count is exactly L/2; M keeps the medians for that case);
//   CTd/CTu counts of bins with value <= tmedian / >= tmedian;
//   R*   for each of the four predicates, the staged position of its r-th true bin, so that "next
//        (previous) bin from x where the predicate holds" -- which is what each of the reference's four
//        trim walks computes (rsi.cpp:1211-1214 / 1241-1244) -- is R[C[x]] (R[C[x+1]-1]): O(1).
// The per-L score test is folded on the host into a threshold on the window sum (hit iff
// sum <= thr_del[L] / sum >= thr_dup[L]).  Every lane owns a bin and walks L = 1..Lmax.  A hit marks
// its trimmed interval [i1,i2] with "smallest L wins" (App. A Q14): instead of touching every bin
// it drops L with atomicMin on the one or two power-of-two blocks that cover the interval (level k =
// floor(log2(length)), capped at kcap, then ceil(length/2^kcap) blocks); after the sweep the levels
// are pushed down to level 0 in kcap steps and the tile's marks are merged into HBM.  A hit costs a
// dozen LDS operations whatever its length; nothing in the hit path leaves LDS.
constexpr int kScanTile = 256;    // one bin per lane: the tile's critical path is one lane's walk over L
constexpr uint32_t kUnmarked = 0xffffffffu;

// IX: the type of a staged index or count -- 16 bits while the staged stretch has fewer than 32 768 bins (every scan whose tile
// fits LDS, and the device-memory tiles up to Lmax = kScanNarrowL), 32 bits for the scans beyond (a computed length above
// 32 000 bins: rsi.cpp:1286-1289 with a large -threshold or a very noisy chromosome)
template <class IX>
struct ScanLdsT {    // plain base pointers and one stride: nothing here is indexed at run time, so it stays in registers
  using index = IX;
  double* P;        // count + 1
  double* tot;      // kThreads
  uint32_t* TB;     // mark levels: [DEL, DUP][kcap + 1 levels][count]
  int* M;           // count: bin medians (the rare straddling median test reads the window)
  IX* CB;           // four prefix-count arrays CL, CG, CTd, CTu, `stride` apart (count + 1 used)
  IX* RB;           // four position-by-rank arrays, `stride` apart
  IX* PF;           // count: predicate bits of each staged bin (bit q = predicate q)
  int stride;       // count + 2
  int count, kcap;
  __device__ IX& C(int q, int x) const { return CB[(size_t)q * stride + x]; }
  __device__ IX& R(int q, int r) const { return RB[(size_t)q * stride + r]; }
  __device__ uint32_t* level(int side, int k) const { return TB + ((size_t)side * (kcap + 1) + k) * count; }
};
enum { kCL = 0, kCG = 1, kCTd = 2, kCTu = 3 };
constexpr int kScanNarrowL = 32000;   // 256 + 2 (Lmax / 2 + 1) staged bins stay below 32 768: 16-bit indices

__host__ __device__ inline size_t scan_lds_bytes(int count, int kcap, int ix_bytes = 2) {
  size_t b = ((size_t)(count + 1) + kThreads) * sizeof(double);
  b += 2 * (size_t)(kcap + 1) * count * 4 + (size_t)count * 4;
  b += 9 * (size_t)(count + 2) * ix_bytes;
  return b;
}
template <class SL>
__device__ inline void scan_lds_carve(SL& S, double* sm, int count, int kcap) {
  S.count = count; S.kcap = kcap; S.stride = count + 2;
  S.P = sm;
  S.tot = S.P + count + 1;
  S.TB = reinterpret_cast<uint32_t*>(S.tot + kThreads);
  S.M = reinterpret_cast<int*>(S.TB + 2 * (size_t)(kcap + 1) * count);
  S.CB = reinterpret_cast<typename SL::index*>(S.M + count);
  S.RB = S.CB + 4 * (size_t)S.stride;
  S.PF = S.RB + 4 * (size_t)S.stride;
}

// first staged position >= x where predicate q holds, vhi when none; last position <= x, vlo-1 when none
template <class SL>
__device__ inline int scan_next(const SL& S, int q, int x, int vhi) {
  const int r = S.C(q, x);
  return r < (int)S.C(q, S.count) ? (int)S.R(q, r) : vhi;
}
template <class SL>
__device__ inline int scan_prev(const SL& S, int q, int x, int vlo) {
  const int r = S.C(q, x + 1);
  return r > 0 ? (int)S.R(q, r - 1) : vlo - 1;
}
template <class SL>
__device__ inline void scan_mark(const SL& S, int side, int lo, int hi, int L) {
  const int len = hi - lo + 1;
  int k = 31 - __clz(len);
  k = k > S.kcap ? S.kcap : k;
  const int step = 1 << k;
  uint32_t* tab = S.level(side, k);
  for (int a = lo; a + step - 1 < hi; a += step) atomicMin(&tab[a], (uint32_t)L);
  atomicMin(&tab[hi - step + 1], (uint32_t)L);
}

// Trim walks and marks of one hit that passed the median test, for the window [w0, w0+L-1] (staged
// indices), when the lane has no run to continue (see scan_sweep).  side 0 = DEL, 1 = DUP; vlo/vhi:
// staged indices inside the chromosome are [vlo, vhi); ends: bit 0 = the tile touches the chromosome
// start, bit 1 = its end.  Returns i1 | i2 << 32 (the trimmed interval, possibly empty), or -1 when a
// walk left the staged range.  Out of line: it is the rare case; it finds the tile's LDS through the
// kernel's dynamic LDS symbol, so every access stays an LDS access.
template <class SL>
__device__ __noinline__ long long scan_hit_slow(int count, int kcap, int w0, int L, int side, int vlo, int vhi, int ends,
                                                uint32_t* counters, double* gtile /* NULL: the tile is the kernel's dynamic LDS */) {
  extern __shared__ __align__(16) double sm[];
  SL S;
  if (gtile) scan_lds_carve(S, gtile, count, kcap); else scan_lds_carve(S, sm, count, kcap);
  const int qm = side ? kCG : kCL, qt = side ? kCTu : kCTd;
  // the four trim walks in the reference's order (rsi.cpp:1211-1214 / 1241-1244), bounded to the chromosome
  int i1 = scan_next(S, qt, w0, vhi);
  i1 = scan_next(S, qm, i1, vhi);
  int i2 = scan_prev(S, qt, w0 + L - 1, vlo);
  i2 = scan_prev(S, qm, i2, vlo);
  if (i1 >= vhi || i2 < vlo) {
    // left the staged range: the marked interval is empty either way.  Leaving the chromosome
    // itself is where the reference aborts (App. A Q12): count those.
    if ((i1 >= vhi && (ends & 2)) || (i2 < vlo && (ends & 1))) atomicAdd(&counters[0], 1u);
    return -1;
  }
  if (i1 <= i2) scan_mark(S, side, i1, i2, L);
  return (long long)i1 | ((long long)i2 << 32);
}

struct ScanTile { int vlo, vhi, fl_del, ce_dup, ends; double lim_del, lim_dup; double* gtile; };

// One lane's walk over L = 1..Lmax in groups of kScanPad (8): the eight prefix values a group needs
// are independent LDS reads issued together, the sixteen thresholds are wave-uniform (scalar)
// loads, so a group costs one memory latency instead of one per L.  Going from L-1 to L the window
// gains one bin -- on the left for even L, on the right for odd L -- so one prefix value per step
// is new.  The score tests of a group set bits; the (rare) set bits are then handled in a rolled
// loop, in increasing L per side.  EDGE: some lanes stop before Lmax (chromosome ends).
//
// Hits come in runs: a bin inside an event hits at every L once its window is long enough, and
// apart from those runs hits are rare (0.5 % of the (bin, L) pairs of a 30x genome, in 0.5 % of the
// waves).  The lane therefore keeps, per side:
//  * ScanRun: the trimmed interval [i1, i2] of its last hit and that hit's L.  When the next hit
//    comes at L+1 the window has gained one bin g; the walk on the other side ends where it did, and
//    on the side of g:
//      value and median predicate hold at g   -> the interval now ends at g;
//      the value predicate fails at g         -> the walk passes over g exactly as before: same end;
//      only the median predicate fails at g   -> the end is the nearest median-predicate bin beyond g
//                                                (one rank lookup; never inside the old interval).
//    The interval only grows, its old part already carries marks <= L, so only the new bins get L.
//    Without a run to continue the hit goes through scan_hit_slow.
//  * ScanMid: for the exact median test when the count says the two middle elements straddle the
//    limit (even L, exactly L/2 bins beyond it): a = the largest median on the near side, b = the
//    smallest on the far side, over the window of length `upto`.  The lane's windows are nested, so
//    the pair is extended by the bins gained since (a bin on an event's edge straddles at every even
//    L: two reads per step instead of L).
struct ScanRun { int lastL, i1, i2; };
struct ScanMid { int upto, a, b; };

// The hits of one side (SIDE 0 = DEL, 1 = DUP) in one group of kScanPad lengths, in increasing L.  What every length of the group
// may need from LDS -- the two ends of the median-predicate count of its window and the predicate bits of the bin the window
// gained -- is read for all eight lengths at once, before the first hit is looked at: inside an event every lane hits at every
// L, and three dependent LDS round trips per hit add up over a hundred hits.  The rare cases (first hit of a run, a bin whose
// median predicate fails, the straddling median) still go to LDS when they come up.
template <int SIDE, class SL>
__device__ inline void scan_hits_side(const SL& S, const ScanTile& t, unsigned hs, int relc, int L0, int Lmax, ScanRun& run,
                                      ScanMid& mid, uint32_t* counters) {
  constexpr int qm = SIDE ? kCG : kCL, qt = SIDE ? kCTu : kCTd;
  int cw[kScanPad];          // bins of the window beyond the median limit
  unsigned pb[kScanPad];     // predicate bits of the bin the window gained at this length
#pragma unroll
  for (int u = 0; u < kScanPad; ++u) {
    int L = L0 + u;
    L = L > Lmax ? Lmax : L;                       // a bit of hs is never set past Lmax; stay in range
    const int hh = L >> 1, w0 = relc - hh;
    cw[u] = (int)S.C(qm, w0 + L) - (int)S.C(qm, w0);
    pb[u] = S.PF[(L & 1) ? w0 + L - 1 : w0];       // even L: the window grew on the left
  }
#pragma unroll
  for (int u = 0; u < kScanPad; ++u) {
    if (!((hs >> u) & 1u)) continue;
    const int L = L0 + u;
    const bool grew_left = !(L & 1);
    const int hh = L >> 1, w0 = relc - hh, e = w0 + L - 1;
    const int c = cw[u];
    bool pass = c >= hh + 1;                         // enough bins beyond the limit for either parity
    if (!pass && grew_left && c == hh) {             // even L, the two middle elements straddle the limit: need their values
      ScanMid m = mid;
      const int thr = SIDE ? t.ce_dup : t.fl_del;
      const int ow0 = relc - (m.upto >> 1), oe = ow0 + m.upto - 1;   // window the pair covers (empty for upto = 0)
      const int* M = S.M;
      // DEL: a = max{x <= thr}, b = min{x > thr}; DUP: a = max{x < thr}, b = min{x >= thr}
      auto fold = [&](int j) {
        const int x = M[j];
        const bool near = SIDE ? x < thr : x <= thr;
        if (near) m.a = x > m.a ? x : m.a; else m.b = x < m.b ? x : m.b;
      };
      for (int j = w0; j < ow0; ++j) fold(j);        // the bins gained on the left since ...
      for (int j = oe + 1; j <= e; ++j) fold(j);     // ... and on the right
      m.upto = L;
      mid = m;
      const double md = 0.5 * ((double)m.a + (double)m.b);
      pass = SIDE ? !(md < t.lim_dup) : !(md > t.lim_del);   // rsi.cpp:1236 / 1206
    }
    const ScanRun old = run;
    ScanRun now = {-1, 0, 0};
    if (pass && old.lastL == L - 1) {                // the run continues
      const int g = grew_left ? w0 : e;
      const unsigned bits = pb[u];
      const bool vt = (bits >> qt) & 1u, vm = (bits >> qm) & 1u;
      now.lastL = L; now.i1 = old.i1; now.i2 = old.i2;
      int lo_m, hi_m;
      if (grew_left) {
        if (vt) now.i1 = vm ? g : scan_next(S, qm, g, t.vhi);
        lo_m = now.i1; hi_m = now.i2 < old.i1 - 1 ? now.i2 : old.i1 - 1;
      } else {
        if (vt) now.i2 = vm ? g : scan_prev(S, qm, g, t.vlo);
        lo_m = now.i1 > old.i2 + 1 ? now.i1 : old.i2 + 1; hi_m = now.i2;
      }
      if (lo_m <= hi_m) scan_mark(S, SIDE, lo_m, hi_m, L);
    } else if (pass) {
      const long long r = scan_hit_slow<SL>(S.count, S.kcap, w0, L, SIDE, t.vlo, t.vhi, t.ends, counters, t.gtile);
      if (r >= 0) { now.lastL = L; now.i1 = (int)(r & 0xffffffffll); now.i2 = (int)(r >> 32); }
    }
    run = now;
  }
}

// [Lbeg, Lfin]: the lengths this workgroup looks at (Lbeg = 1 mod kScanPad, whole groups): a tile's lengths are split over
// several workgroups (k_rsi_scan).  A lane that starts in the middle has no run to continue: its first hit takes the general
// path (scan_hit_slow), exactly as the first hit of a run does, and smallest-L-wins is an atomicMin whoever comes first.
template <bool EDGE, class SL>
__device__ inline void scan_sweep(const SL& S, const ScanTile& t, const double* __restrict__ thr_del,
                                  const double* __restrict__ thr_dup, int relc, int Lmax, int Lend, int Lbeg, int Lfin, uint32_t* counters) {
  double p_lo = S.P[relc - ((Lbeg - 1) >> 1)], p_hi = 0.0;   // the left end of the window of length Lbeg - 1
  ScanRun run_del = {-1, 0, 0}, run_dup = {-1, 0, 0};
  ScanMid mid_del = {0, (int)0x80000000, 0x7fffffff}, mid_dup = {0, (int)0x80000000, 0x7fffffff};
  for (int L0 = Lbeg; L0 <= Lfin; L0 += kScanPad) {
    double pv[kScanPad], td[kScanPad], tu[kScanPad];
#pragma unroll
    for (int u = 0; u < kScanPad; ++u) {
      int L = L0 + u;
      L = L > Lmax ? Lmax : L;                      // past Lmax the thresholds are unreachable; just stay in range
      const int h = L >> 1;
      pv[u] = S.P[(u & 1) ? relc - h : relc + h + 1];   // L0 is odd: odd u <=> even L <=> the window grew on the left
      td[u] = thr_del[L0 + u];
      tu[u] = thr_dup[L0 + u];
    }
    unsigned hits = 0;   // bit u: DEL score hit at L0+u, bit 8+u: DUP
#pragma unroll
    for (int u = 0; u < kScanPad; ++u) {
      if (u & 1) p_lo = pv[u]; else p_hi = pv[u];
      const double sum = p_hi - p_lo;
      bool hd = sum <= td[u], hu = sum >= tu[u];
      if (EDGE) { const bool live = L0 + u <= Lend; hd = hd && live; hu = hu && live; }
      hits |= (hd ? 1u << u : 0u) | (hu ? 0x100u << u : 0u);
    }
    if (!__ballot(hits != 0)) continue;
    if (__ballot((hits & 0xffu) != 0)) scan_hits_side<0, SL>(S, t, hits & 0xffu, relc, L0, Lmax, run_del, mid_del, counters);
    if (__ballot((hits >> 8) != 0)) scan_hits_side<1, SL>(S, t, hits >> 8, relc, L0, Lmax, run_dup, mid_dup, counters);
  }
}

// ---- K7a  scan_detect: which tiles can have a hit at all? ------------------------------------------------------------------
// Nearly every (bin, L) pair is far from both thresholds: 0.5 % of the pairs of a 30x genome are hits, all of them in the few
// tiles that touch an event -- and those tiles are expensive: every lane inside an event hits at every L and pays the whole hit
// path a hundred times in sequence, so with one kernel for everything a single wave that lay in an event was the critical
// path of the launch (a fixed 70 us of a 150 us pass, whatever the chromosome's length and however few its events), and every
// tile paid for staging the structures only hits need (mark tables, rank arrays, bin medians).
// So the scan is two launches.  This one decides, tile by tile, whether any lane can hit at any L, from FLOAT prefixes:
// |fl32(Pf[hi] - Pf[lo]) - (P[hi] - P[lo])| <= 3 * 2^-24 * max|P| (one rounding per prefix -- they are rounded from the exact
// double prefixes -- and one in the subtraction), so against thresholds widened by 4 * 2^-24 * max|P| of the workgroup's own
// stretch and rounded outwards, "no float hit" implies "no exact hit".  It lists the tiles where a float test fires; k_rsi_scan
// then runs the exact sweep on those tiles only, its lengths split over several workgroups.  A listed tile without a real hit
// costs time, never correctness; a chromosome-end tile ignores the cut-off of its lanes' lengths here (a superset again).
// One workgroup = kDetTiles tiles of kScanTile bins (one per wave; lane l looks at bins l, l + 64, l + 128, l + 192 of its
// tile: four independent chains per lane), staged with one halo for all of them.
constexpr int kScanPartsMax = 8;                   // shares of a listed tile's lengths (k_rsi_scan)
constexpr int kDetTiles = 4;                       // tiles per workgroup = waves per workgroup
constexpr int kDetBins = kDetTiles * kScanTile;    // 1024
__host__ __device__ inline size_t detect_lds_bytes(int Lmax) {
  const int halo = Lmax / 2 + 1, count = kDetBins + 2 * halo;
  return (size_t)(Lmax + 1 + kScanPad) * 8 + (((size_t)count + 2) & ~(size_t)1) * 4 + 16;
}
template <bool INL>
__global__ __launch_bounds__(kThreads) void k_scan_detect(const float* __restrict__ T, int64_t nb, int Lmax,
                                                          const double* __restrict__ thr_del_mem, const double* __restrict__ thr_dup_mem,
                                                          uint32_t* __restrict__ tiles /* out: listed tiles */, uint32_t* __restrict__ counters
                                                          /* [1]: inexact values, [8]: tiles listed */, ScanThr inl) {
  extern __shared__ __align__(16) double sm[];
  const double* __restrict__ thr_del = INL ? inl.del : thr_del_mem;
  const double* __restrict__ thr_dup = INL ? inl.dup : thr_dup_mem;
  const int halo = Lmax / 2 + 1;
  const int count = kDetBins + 2 * halo;
  float2* TH = reinterpret_cast<float2*>(sm);                           // per L: (DEL, DUP) float thresholds of this workgroup
  float* Pf = reinterpret_cast<float*>(TH + (Lmax + 1 + kScanPad));     // count + 1 prefixes
  __shared__ double s_tot[kThreads];
  __shared__ unsigned int s_pmax;
  const int64_t first = (int64_t)blockIdx.x * kDetBins;
  const int64_t lo = first - halo;
  if (threadIdx.x == 0) s_pmax = 0u;
  // ---- exact double prefix of the staged stretch (serial chunk per thread + scan of the chunk totals), rounded to float ----
  const int chunk = (count + kThreads - 1) / kThreads;
  const int c0 = threadIdx.x * chunk;
  double run = 0.0;
  unsigned int inexact = 0;
  for (int e = c0; e < c0 + chunk && e < count; ++e) {
    const int64_t i = lo + e;
    const float v = (i >= 0 && i < nb) ? T[i] : 0.0f;
    // exact-sum precondition of the exact sweep: 0, or 2^-10 <= |v| < 2^20; counted once, by the owning workgroup
    const float av = fabsf(v);
    if (!(av == 0.0f || (av >= 0.0009765625f && av < 1048576.0f)) && e >= halo && e < halo + kDetBins) inexact++;
    run += (double)v;
  }
  s_tot[threadIdx.x] = run;
  __syncthreads();
  if (threadIdx.x < 64) {   // wave 0 turns the 256 totals into exclusive offsets
    double carry = 0.0;
    for (int k = 0; k < kThreads / 64; ++k) {
      const int idx = k * 64 + threadIdx.x;
      const double mine = s_tot[idx];
      const double incl = wave_incl_scan(mine);   // DPP: the sums are exact in double, the association does not matter
      s_tot[idx] = carry + incl - mine;
      carry += wave_lane63(incl);
    }
  }
  __syncthreads();
  {
    double acc = s_tot[threadIdx.x];
    float amax = 0.0f;
    if (threadIdx.x == 0) Pf[0] = 0.0f;
    for (int e = c0; e < c0 + chunk && e < count; ++e) {
      const int64_t i = lo + e;
      const float v = (i >= 0 && i < nb) ? T[i] : 0.0f;   // a second read (L1 / L2) instead of 8 bytes of LDS per staged bin
      acc += (double)v;
      const float pf = (float)acc;
      Pf[e + 1] = pf;
      const float a = fabsf(pf);
      amax = (a > amax || !(a == a)) ? a : amax;          // a NaN sticks: the thresholds then list every tile of the workgroup
    }
    // max over the stretch as an integer maximum of float bits (the order of non-negative floats; NaN and infinity on top)
    unsigned int ab = __float_as_uint(amax);
    for (int d = 32; d >= 1; d >>= 1) { const unsigned int o = (unsigned int)__shfl_xor((int)ab, d); ab = o > ab ? o : ab; }
    if (lane_id() == 0) atomicMax(&s_pmax, ab);
  }
  for (int d = 32; d >= 1; d >>= 1) inexact += __shfl_xor(inexact, d);
  if (lane_id() == 0 && inexact) atomicAdd(&counters[1], inexact);
  __syncthreads();
  {
    const double pmax = (double)__uint_as_float(s_pmax);
    const double margin = pmax * (4.0 / 16777216.0);
    const bool open_all = !(margin == margin) || margin > 3.0e38;   // NaN / overflowing bound: every sum is a candidate
    for (int L = threadIdx.x; L < Lmax + 1 + kScanPad; L += kThreads) {
      const double d = thr_del[L] + margin, u = thr_dup[L] - margin;
      float fd = (float)d, fu = (float)u;
      if ((double)fd < d) fd = nextafterf(fd, INFINITY);
      if ((double)fu > u) fu = nextafterf(fu, -INFINITY);
      if (open_all) { fd = INFINITY; fu = -INFINITY; }
      TH[L] = make_float2(fd, fu);
    }
  }
  __syncthreads();
  // ---- every (bin, L) of the wave's tile, four bins per lane; the wave stops at its first candidate ----
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int64_t tile = (int64_t)blockIdx.x * kDetTiles + wave;
  if (tile * kScanTile >= nb) return;
  const int rel0 = halo + wave * kScanTile + lane;          // staged index of the lane's first bin
  float f_lo[4], f_hi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int b = 0; b < 4; ++b) f_lo[b] = Pf[rel0 + 64 * b];
  unsigned long long cand = 0;
  for (int L0 = 1; L0 <= Lmax && cand == 0; L0 += kScanPad) {
    float2 th[kScanPad];
    int off[kScanPad];   // wave-uniform: staged offset of the prefix value that is new at this length
#pragma unroll
    for (int u = 0; u < kScanPad; ++u) {
      int L = L0 + u;
      L = L > Lmax ? Lmax : L;                              // past Lmax the thresholds are unreachable; just stay in range
      const int h = L >> 1;
      off[u] = (u & 1) ? -h : h + 1;                        // L0 is odd: odd u <=> even L <=> the window grew on the left
      th[u] = TH[L0 + u];                                   // the same address in every lane: a broadcast read
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float pf[kScanPad];
#pragma unroll
      for (int u = 0; u < kScanPad; ++u) pf[u] = Pf[rel0 + 64 * b + off[u]];
#pragma unroll
      for (int u = 0; u < kScanPad; ++u) {
        if (u & 1) f_lo[b] = pf[u]; else f_hi[b] = pf[u];
        const float fs = f_hi[b] - f_lo[b];
        // wave masks straight from the compares, OR-ed with scalar instructions: no per-lane flag, no branches
        cand |= __builtin_amdgcn_ballot_w64(fs <= th[u].x) | __builtin_amdgcn_ballot_w64(fs >= th[u].y);
      }
    }
  }
  if (cand != 0 && lane == 0) tiles[atomicAdd(&counters[8], 1u)] = (uint32_t)tile;
}

// ---- K7b  rsi_scan: the exact sweep on the listed tiles -----------------------------------------------------------------
// INL: the thresholds ride in the kernel arguments (Lmax <= kThrInline - kScanPad - 1: every configuration BASELINE names);
// otherwise they are read from device memory the host uploaded.
// A task = (listed tile, share of the lengths): `parts` tasks per tile, each staging the tile for itself and sweeping its whole
// groups of kScanPad lengths -- the chain of hits a lane inside an event walks through is cut in `parts`, the tile's marks are
// atomicMin whoever makes them.  Workgroups pull tasks from a counter (counters[9]) until it runs past the list; tiles ==
// NULL: every tile of the chromosome is a task list of its own (no detection pass in front).
// GTILE: the tile lives in device memory (gws + blockIdx.x * gbytes) instead of LDS -- scans longer than kScanLdsL.  Same code;
// the waves of a workgroup share a CU and its L1, so what one wave stored is what another loads once every wave's stores are
// done (drain) and the workgroup has met at the barrier.
// WIDE (device-memory tiles only): 32-bit staged indices, for the scans whose staged stretch has 32 768 bins and more.
template <bool INL, bool GTILE, bool WIDE>
__global__ __launch_bounds__(kThreads) void k_rsi_scan(const float* __restrict__ T, const int32_t* __restrict__ medint,
                                                       ScanParams sp, const double* __restrict__ thr_del_mem,
                                                       const double* __restrict__ thr_dup_mem,
                                                       uint32_t* __restrict__ first_del, uint32_t* __restrict__ first_dup,
                                                       uint32_t* __restrict__ counters, const uint32_t* __restrict__ tiles, int parts,
                                                       ScanThr inl, unsigned char* __restrict__ gws, unsigned long long gbytes) {
  extern __shared__ __align__(16) double sm[];
  const double* __restrict__ thr_del = INL ? inl.del : thr_del_mem;
  const double* __restrict__ thr_dup = INL ? inl.dup : thr_dup_mem;
  const int Lmax = sp.Lmax, kcap = sp.kcap;
  const int halo = Lmax / 2 + 1;
  const int count = kScanTile + 2 * halo;          // staged bins
  auto tile_sync = [&]() { if (GTILE) drain(); __syncthreads(); };
  static_assert(GTILE || !WIDE, "a tile with 32-bit indices lives in device memory");
  using IX = typename std::conditional<WIDE, uint32_t, uint16_t>::type;
  using SL = ScanLdsT<IX>;
  SL S;
  double* const gtile = GTILE ? reinterpret_cast<double*>(gws + (size_t)blockIdx.x * gbytes) : nullptr;
  if (GTILE) scan_lds_carve(S, gtile, count, kcap); else scan_lds_carve(S, sm, count, kcap);
  __shared__ int s_c[4][kThreads];
  __shared__ unsigned int s_task;
  // integer forms of the median limits: x > lim_del <=> x > fl_del ; x < lim_dup <=> x < ce_dup
  const int fl_del = (int)floor(sp.lim_del), ce_dup = (int)ceil(sp.lim_dup);
  const double tmed = sp.tmedian;
  const int groups = (Lmax + kScanPad - 1) / kScanPad, per_part = (groups + parts - 1) / parts;
  const unsigned int ntiles = tiles ? counters[8] : (unsigned int)((sp.nb + kScanTile - 1) / kScanTile);
  const unsigned int ntasks = ntiles * (unsigned int)parts;
  for (;;) {
    tile_sync();   // the previous task's LDS is no longer read
    if (threadIdx.x == 0) s_task = atomicAdd(&counters[9], 1u);
    tile_sync();
    const unsigned int task = s_task;
    if (task >= ntasks) break;   // every wave reaches this: the grid drains
    const unsigned int tslot = task / (unsigned int)parts;
    const int part = (int)(task - tslot * (unsigned int)parts);
    const int Lbeg = 1 + part * per_part * kScanPad;
    int Lfin = (part + 1) * per_part * kScanPad;
    Lfin = Lfin > Lmax ? Lmax : Lfin;
    if (Lbeg > Lfin) continue;
    const int64_t tile_start = (int64_t)(tiles ? tiles[tslot] : tslot) * kScanTile;
    const int64_t lo = tile_start - halo;
    const int vlo = lo < 0 ? (int)(-lo) : 0;
    const int vhi = (lo + count > sp.nb) ? (int)(sp.nb - lo) : count;
    const bool at_start = lo <= 0, at_end = lo + count >= sp.nb;
    for (int e = threadIdx.x; e < 2 * (kcap + 1) * count; e += kThreads) S.TB[e] = kUnmarked;
    // ---- stage + exact prefixes: serial chunk per thread, then a scan of the 256 chunk totals ----
    const int chunk = (count + kThreads - 1) / kThreads;
    const int c0 = threadIdx.x * chunk;
    double run = 0.0;
    int rc[4] = {0, 0, 0, 0};
    unsigned int inexact = 0;
    for (int e = c0; e < c0 + chunk && e < count; ++e) {
      const bool in = e >= vlo && e < vhi;
      const float v = in ? T[lo + e] : 0.0f;
      const int mi = in ? medint[lo + e] : 0;
      // exact-sum precondition: 0, or 2^-10 <= |v| < 2^20; counted once per bin: by the detection pass
      // when there is one, else by the owning tile's first part
      const float av = fabsf(v);
      if (!tiles && part == 0 && !(av == 0.0f || (av >= 0.0009765625f && av < 1048576.0f)) && e >= halo && e < halo + kScanTile) inexact++;
      S.M[e] = mi;
      run += (double)v;
      rc[kCL] += (in && mi <= fl_del);
      rc[kCG] += (in && mi >= ce_dup);
      rc[kCTd] += (in && !((double)v > tmed));       // where a DEL value walk stops (rsi.cpp:1211, 1213)
      rc[kCTu] += (in && !((double)v < tmed));       // where a DUP value walk stops (rsi.cpp:1241, 1243)
      S.P[e + 1] = run;
#pragma unroll
      for (int q = 0; q < 4; ++q) S.C(q, e + 1) = (IX)rc[q];
    }
    S.tot[threadIdx.x] = run;
#pragma unroll
    for (int q = 0; q < 4; ++q) s_c[q][threadIdx.x] = rc[q];
    if (threadIdx.x == 0) { S.P[0] = 0.0; for (int q = 0; q < 4; ++q) S.C(q, 0) = 0; }
    tile_sync();
    if (threadIdx.x < 64) {   // wave 0 turns the 256 totals into exclusive offsets
      double carry = 0.0;
      int car[4] = {0, 0, 0, 0};
      for (int k = 0; k < kThreads / 64; ++k) {
        const int idx = k * 64 + threadIdx.x;
        const double mine = S.tot[idx];
        const double incl = wave_incl_scan(mine);   // DPP row shifts and broadcasts: no trips through the LDS crossbar
        int m4[4], i4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { m4[q] = s_c[q][idx]; i4[q] = wave_incl_scan(m4[q]); }
        S.tot[idx] = carry + incl - mine;
        carry += wave_lane63(incl);
#pragma unroll
        for (int q = 0; q < 4; ++q) { s_c[q][idx] = car[q] + i4[q] - m4[q]; car[q] += wave_lane63(i4[q]); }
      }
    }
    tile_sync();
    {
      const double off = S.tot[threadIdx.x];
      int o4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o4[q] = s_c[q][threadIdx.x];
      int before[4] = {o4[0], o4[1], o4[2], o4[3]};   // global count before element e
      for (int e = c0; e < c0 + chunk && e < count; ++e) {
        S.P[e + 1] += off;
        unsigned bits = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int after = (int)S.C(q, e + 1) + o4[q];
          S.C(q, e + 1) = (IX)after;
          if (after != before[q]) { S.R(q, before[q]) = (IX)e; bits |= 1u << q; }   // e is the before[q]-th bin with predicate q
          before[q] = after;
        }
        S.PF[e] = (IX)bits;
      }
    }
    for (int d = 32; d >= 1; d >>= 1) inexact += __shfl_xor(inexact, d);
    if (lane_id() == 0 && inexact) atomicAdd(&counters[1], inexact);
    tile_sync();

    // ---- every (bin, L) of the tile within this task's lengths.  Going from L-1 to L the window gains one bin -- on the left for
    // even L, on the right for odd L -- so one prefix value per step is new.  ----
    ScanTile tile;
    tile.vlo = vlo; tile.vhi = vhi; tile.fl_del = fl_del; tile.ce_dup = ce_dup;
    tile.lim_del = sp.lim_del; tile.lim_dup = sp.lim_dup; tile.ends = (at_start ? 1 : 0) | (at_end ? 2 : 0);
    tile.gtile = gtile;
    const bool edge_tile = at_start || at_end;   // only there can a lane's L range be cut short
    static_assert(kScanTile == kThreads, "one bin per lane");
    {
      const int64_t i = tile_start + threadIdx.x;
      const int rel = (int)(i - lo);   // staged index of bin i
      // the reference visits i in [L/2+1, nb-L/2-2] (rsi.cpp:1204): L/2 <= min(i-1, nb-i-2)
      const int64_t hmax = (i - 1) < (sp.nb - i - 2) ? (i - 1) : (sp.nb - i - 2);
      int Lend = (i < sp.nb && hmax >= 0) ? (int)(hmax < Lmax ? 2 * hmax + 1 : Lmax) : 0;
      if (Lend > Lmax) Lend = Lmax;
      const int relc = Lend > 0 ? rel : halo;   // lanes without a bin read a harmless address
      if (edge_tile) scan_sweep<true, SL>(S, tile, thr_del, thr_dup, relc, Lmax, Lend, Lbeg, Lfin, counters);
      else scan_sweep<false, SL>(S, tile, thr_del, thr_dup, relc, Lmax, Lend, Lbeg, Lfin, counters);
    }
    tile_sync();
    // ---- push the block levels down to single bins ----
    for (int k = kcap; k >= 1; --k) {
      const int half = 1 << (k - 1);
      uint32_t* hd = S.level(0, k); uint32_t* ld = S.level(0, k - 1);
      uint32_t* hu = S.level(1, k); uint32_t* lu = S.level(1, k - 1);
      for (int e = threadIdx.x; e < count; e += kThreads) {
        const uint32_t d = hd[e], u = hu[e];
        if (d != kUnmarked) { atomicMin(&ld[e], d); atomicMin(&ld[e + half], d); }
        if (u != kUnmarked) { atomicMin(&lu[e], u); atomicMin(&lu[e + half], u); }
      }
      tile_sync();
    }
    // ---- merge the tile's marks into HBM (halo bins are shared with the neighbouring tiles and with the tile's other parts) ----
    for (int e = vlo + threadIdx.x; e < vhi; e += kThreads) {
      const uint32_t d = S.level(0, 0)[e], u = S.level(1, 0)[e];
      if (d != kUnmarked) atomicMin(&first_del[lo + e], d);
      if (u != kUnmarked) atomicMin(&first_dup[lo + e], u);
    }
  }
}

// After a sweep pair: at which L did each sweep stop?  The DEL sweep stops after the first L with more than a fifth of
// the bins marked (rsi.cpp:1225), the DUP sweep likewise (rsi.cpp:1255) but it cannot mark bins the DEL sweep marked
// (App. A Q14).  One launch: every workgroup histograms first_del and first_dup over L in LDS and appends the (rare) bins
// marked by both sweeps to a list; the last workgroup to finish walks the DEL histogram to its stop level, takes the
// listed bins the DEL sweep really marked out of the DUP histogram (or, if the list overflowed, recounts them itself),
// walks that one, and leaves the levels, the per-L counts (what the reference logs per L, rsi.cpp:1221-1224, 1251-1254)
// and the pass counters in `work`, a copy of which goes to mapped host memory.
// work: [ScanPassHead (64 bytes)] [hist_del] [hist_dup], scan_level_stride(Lmax) words each
constexpr int kBothCap = 16384;
// The first L whose cumulated share of the bins exceeds 0.2 (rsi.cpp:1217-1225), or Lmax: the whole workgroup calls it and gets the
// same answer.  256 levels at a time -- a block scan of their counts, the reference's own expression per level, the smallest
// level that passes -- where one thread used to walk the levels with a double division each (a hundred of them: 6 us per call,
// two calls per launch).
template <class Load>
__device__ inline uint32_t stop_level_block(Load count_at /* count_at(L): bins first marked at length L */, int32_t Lmax, int64_t nb) {
  __shared__ int s_wsum[kThreads / 64], s_wmin[kThreads / 64];
  long long carry = 0;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  for (int base = 1; base <= Lmax; base += kThreads) {
    const int L = base + (int)threadIdx.x;
    const int c = L <= Lmax ? (int)count_at(L) : 0;
    const int incl = wave_incl_scan(c);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    long long before = carry, total = 0;
    for (int w = 0; w < kThreads / 64; ++w) { if (w < wave) before += s_wsum[w]; total += s_wsum[w]; }
    const long long cum = before + incl;
    int first = (L <= Lmax && (double)(int)cum / (double)(int)nb > 0.2) ? L : 0x7fffffff;
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(first, d); first = o < first ? o : first; }
    if (lane == 0) s_wmin[wave] = first;
    __syncthreads();
    int best = 0x7fffffff;
    for (int w = 0; w < kThreads / 64; ++w) best = s_wmin[w] < best ? s_wmin[w] : best;
    __syncthreads();   // s_wsum / s_wmin are rewritten by the next round (and by the next call)
    if (best != 0x7fffffff) return (uint32_t)best;
    carry += total;
  }
  return (uint32_t)Lmax;
}
// WIDE_L: the two per-L histograms do not fit LDS (Lmax beyond 20 000): every count goes straight to the histograms in device memory
// (atomics), and the last workgroup walks them there.  Same results; the scans that long are slow elsewhere.
template <bool WIDE_L>
__global__ __launch_bounds__(kThreads) void k_level_stop(const uint32_t* __restrict__ first_del, const uint32_t* __restrict__ first_dup,
                                                         int64_t nb, int32_t Lmax, uint32_t* __restrict__ work,
                                                         uint2* __restrict__ both, unsigned int* __restrict__ counter,
                                                         void* host_copy, unsigned int host_bytes) {
  extern __shared__ unsigned int s_l[];   // [2][Lmax + 1]
  unsigned int* s_d = s_l;
  unsigned int* s_u = s_l + (Lmax + 1);
  uint32_t* head = work;                  // [0] escapes [1] inexact (the scan) [2] ldel [3] ldup [4] both-count [5] last run start + 1
  uint32_t* hist_d = work + 16;
  uint32_t* hist_u = hist_d + ((Lmax + 1 + kScanPad + 3) & ~3);   // scan_level_stride(Lmax)
  if (WIDE_L) { s_d = hist_d; s_u = hist_u; }
  else {
    for (int e = threadIdx.x; e < 2 * (Lmax + 1); e += kThreads) s_l[e] = 0;
    __syncthreads();
  }
  auto one = [&](uint32_t fd, uint32_t fu) {
    const bool hd = fd <= (uint32_t)Lmax, hu = fu <= (uint32_t)Lmax;
    if (hd) atomicAdd(&s_d[fd], 1u);
    if (hu) atomicAdd(&s_u[fu], 1u);
    if (hd && hu) { const uint32_t k = atomicAdd(&head[4], 1u); if (k < (uint32_t)kBothCap) { st_cg(&both[k].x, fd); st_cg(&both[k].y, fu); } }
  };
  {   // four of a thread's strided bins per trip: eight loads in flight instead of two
    const int64_t stride = (int64_t)gridDim.x * kThreads;
    int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    for (; i + 3 * stride < nb; i += 4 * stride) {
      uint32_t fd[4], fu[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { fd[e] = first_del[i + e * stride]; fu[e] = first_dup[i + e * stride]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) one(fd[e], fu[e]);
    }
    for (; i < nb; i += stride) one(first_del[i], first_dup[i]);
  }
  if (!WIDE_L) {
    __syncthreads();
    for (int e = threadIdx.x; e <= Lmax; e += kThreads) {
      const unsigned int c = s_d[e], u = s_u[e];
      if (c) atomicAdd(&hist_d[e], c);
      if (u) atomicAdd(&hist_u[e], u);
    }
  }
  if (!last_block_done(counter)) return;
  if (!WIDE_L) {
    for (int e = threadIdx.x; e <= Lmax; e += kThreads) { s_d[e] = ld_cg(hist_d + e); s_u[e] = ld_cg(hist_u + e); }
    __syncthreads();
  }
  // (WIDE_L: the counts are read where the atomics left them -- ld_cg, past this CU's cache)
  auto del_at = [&](int L) { return WIDE_L ? ld_cg(hist_d + L) : s_d[L]; };
  auto dup_at = [&](int L) { return WIDE_L ? ld_cg(hist_u + L) : s_u[L]; };
  const uint32_t ldel = stop_level_block(del_at, Lmax, nb);
  const uint32_t nboth = ld_cg(&head[4]);
  if (nboth <= (uint32_t)kBothCap) {
    for (uint32_t k = threadIdx.x; k < nboth; k += kThreads) {
      const uint32_t fd = ld_cg(&both[k].x), fu = ld_cg(&both[k].y);
      if (fd <= ldel) atomicSub(&s_u[fu], 1u);
    }
  } else {   // the list overflowed: this workgroup recounts the DUP histogram with the exclusion
    for (int e = threadIdx.x; e <= Lmax; e += kThreads) { if (WIDE_L) st_cg(&hist_u[e], 0u); else s_u[e] = 0; }
    if (WIDE_L) drain();
    __syncthreads();
    for (int64_t i = threadIdx.x; i < nb; i += kThreads) {
      const uint32_t fu = first_dup[i];
      if (fu <= (uint32_t)Lmax && !(first_del[i] <= ldel)) atomicAdd(&s_u[fu], 1u);
    }
  }
  if (WIDE_L) drain();   // this workgroup's atomics have landed before any of its threads reads the counts again
  __syncthreads();
  if (!WIDE_L) for (int e = threadIdx.x; e <= Lmax; e += kThreads) st_cg(&hist_u[e], s_u[e]);
  const uint32_t ldup = stop_level_block(dup_at, Lmax, nb);
  if (threadIdx.x == 0) { st_cg(&head[2], ldel); st_cg(&head[3], ldup); }
  sync_drained();
  export_words(host_copy, work, host_bytes);
}

// status[j] = -first_del[j] if first_del[j] <= ldel; else +first_dup[j] if <= ldup; else 0 (copy: a second array that
// receives the same values, for filterstatus to trim).  While the values pass through: the boundaries of the marked runs
// (get_continuous_segments with d = 1, rsi.cpp:291-327: a run is a maximal stretch of adjacent bins of one sign), appended
// unordered as (pos << 1 | is_end); the last workgroup hands the count and the first entries to the host.
__device__ inline int resolved_status(uint32_t fd, uint32_t fu, uint32_t ldel, uint32_t ldup) {
  return fd <= ldel ? -(int)fd : (fu <= ldup ? (int)fu : 0);
}
__global__ __launch_bounds__(kThreads) void k_resolve_runs(const uint32_t* __restrict__ first_del, const uint32_t* __restrict__ first_dup,
                                                           const uint32_t* __restrict__ levels, int64_t nb,
                                                           int32_t* __restrict__ status, int32_t* __restrict__ copy,
                                                           uint64_t* __restrict__ runs, uint32_t* __restrict__ count, uint32_t cap,
                                                           unsigned int* __restrict__ counter, void* host_copy, unsigned int host_entries) {
  const uint32_t ldel = levels[0], ldup = levels[1];
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nb; i += (int64_t)gridDim.x * kThreads) {
    const int s = resolved_status(first_del[i], first_dup[i], ldel, ldup);
    status[i] = s;
    if (copy) copy[i] = s;
    if (s == 0) continue;
    const int p = i > 0 ? resolved_status(first_del[i - 1], first_dup[i - 1], ldel, ldup) : 0;
    const int q = i + 1 < nb ? resolved_status(first_del[i + 1], first_dup[i + 1], ldel, ldup) : 0;
    const bool is_start = p == 0 || ((p > 0) != (s > 0));
    const bool is_end = q == 0 || ((q > 0) != (s > 0));
    if (is_start) { const uint32_t k = atomicAdd(count, 1u); if (k < cap) st_cg(reinterpret_cast<unsigned long long*>(&runs[k]), (unsigned long long)i << 1); }
    if (is_end) { const uint32_t k = atomicAdd(count, 1u); if (k < cap) st_cg(reinterpret_cast<unsigned long long*>(&runs[k]), ((unsigned long long)i << 1) | 1ull); }
  }
  if (!host_copy || !last_block_done(counter)) return;
  // [count, pad][entries ...]
  unsigned int* d = static_cast<unsigned int*>(host_copy);
  const uint32_t n = ld_cg(count);
  if (threadIdx.x == 0) { d[0] = n; d[1] = 0; }
  const uint32_t k = n < host_entries ? n : host_entries;
  export_words(d + 2, runs, (size_t)k * 8);
}

// edge trimming of filterstatus (rsi.cpp:1023-1044): one thread per run, runs are disjoint
__global__ __launch_bounds__(kThreads) void k_trim_runs(const float* __restrict__ T, int32_t* __restrict__ status,
                                                        const int32_t* __restrict__ run_start,
                                                        const int32_t* __restrict__ run_end, int nruns, double delthr,
                                                        double addthr, RunsInline inl) {
  // One WORKGROUP per run.  The reference walks inwards from both ends of a run while the value is within the thresholds
  // (rsi.cpp:1029-1043); with one thread per run and one dependent load per step a -MED genome, whose runs are trimmed
  // thousands of bins deep, spent 17.6 ms in this kernel.  The walks are "first position where the predicate fails":
  // 256 positions per step, a workgroup minimum, then the prefix / suffix is cleared in parallel.
  //   left : positions s, s+1, ... while the predicate holds, never position e unless s == e  -> first kept position L
  //   right: positions e, e-1, ... while it holds, stopping above L (e itself is always looked at)
  __shared__ int s_min[kThreads / 64];
  const int r = blockIdx.x;
  if (r >= nruns) return;
  const int s = run_start ? run_start[r] : inl.se[2 * r], e = run_start ? run_end[r] : inl.se[2 * r + 1];
  auto pred = [&](int i) { const int st = status[i]; const double t = (double)T[i]; return (t > delthr && st < 0) || (t < addthr && st > 0); };
  auto first_failure = [&](int origin, int dir, int limit) {   // number of leading positions origin, origin + dir, ... (at most limit) that satisfy pred
    int done = 0;
    while (done < limit) {
      const int i = done + (int)threadIdx.x;
      int f = (i < limit && pred(origin + dir * i)) ? 0x7fffffff : i;
      for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(f, d); f = o < f ? o : f; }
      __syncthreads();
      if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = f;
      __syncthreads();
      f = s_min[0];
      for (int w = 1; w < kThreads / 64; ++w) f = s_min[w] < f ? s_min[w] : f;
      if (f != 0x7fffffff) return f < limit ? f : limit;
      done += kThreads;
    }
    return limit;
  };
  const int lim_left = s < e ? e - s : 1;
  const int zl = first_failure(s, +1, lim_left);
  const int L = s + zl;
  int zr = 0;
  if (s < e) zr = first_failure(e, -1, e - L > 1 ? e - L : 1);   // reads positions above L (and e): none of them is cleared by the left walk
  __syncthreads();
  for (int i = threadIdx.x; i < zl; i += kThreads) status[s + i] = 0;
  for (int i = threadIdx.x; i < zr; i += kThreads) status[e - i] = 0;
}

// ------------------------------------------------------------------------------------------
// K10  max-score sub-segment of each run (get_rsi_segments, rsi.cpp:1060-1117).
// Step 1: exact double prefix of the run's values into scratch (one workgroup per run).
__global__ __launch_bounds__(kThreads) void k_run_prefix(const float* __restrict__ T, const int32_t* __restrict__ run_start,
                                                         const int32_t* __restrict__ run_end,
                                                         const int64_t* __restrict__ poff, double* __restrict__ scratch, RunsInline inl,
                                                         const int32_t* __restrict__ status, int32_t* __restrict__ status_out) {
  __shared__ double s_tot[kThreads];
  const int r = blockIdx.x;
  const int s = run_start ? run_start[r] : inl.se[2 * r];
  const int len = (run_start ? run_end[r] : inl.se[2 * r + 1]) - s + 1;
  const int64_t off = poff ? poff[r] : (int64_t)inl.off[r];
  double* P = scratch + off;
  // the run's status values, run after run (the prefixes take len + 1 slots per run, the values len): the host needs the
  // status array inside the runs only (type of a segment, nested levels), so this replaces a copy of the whole array
  if (status_out) for (int e = threadIdx.x; e < len; e += kThreads) status_out[off - r + e] = status[s + e];
  const int chunk = (len + kThreads - 1) / kThreads;
  const int c0 = threadIdx.x * chunk;
  double run = 0.0;
  for (int e = c0; e < c0 + chunk && e < len; ++e) run += (double)T[s + e];
  s_tot[threadIdx.x] = run;
  __syncthreads();
  if (threadIdx.x < 64) {
    double carry = 0.0;
    for (int k = 0; k < kThreads / 64; ++k) {
      const double mine = s_tot[k * 64 + threadIdx.x];
      const double incl = wave_incl_scan(mine);
      s_tot[k * 64 + threadIdx.x] = carry + incl - mine;
      carry += wave_lane63(incl);
    }
  }
  __syncthreads();
  run = s_tot[threadIdx.x];
  if (threadIdx.x == 0) P[0] = 0.0;
  for (int e = c0; e < c0 + chunk && e < len; ++e) { run += (double)T[s + e]; P[e + 1] = run; }
}

// Step 2: work items (run, Lbeg, Lend); every (L, offset) pair scored as the reference does,
// best kept under the reference's visiting order: larger score, then smaller L, then smaller offset.
__device__ inline bool seg_better(double s, int L, int j, double bs, int bL, int bj) {
  if (s != bs) return s > bs;
  if (L != bL) return L < bL;
  return j < bj;
}

constexpr int kBestLds = 6144;   // doubles of a run's prefix staged in LDS (48 KB)
__global__ __launch_bounds__(kThreads) void k_best_subsegment(const SegItem* __restrict__ items,
                                                              const int64_t* __restrict__ poff,
                                                              const double* __restrict__ scratch, double tmedian,
                                                              BestSeg* __restrict__ out, ItemsInline inl) {
  __shared__ double s_s[kThreads];
  __shared__ int s_L[kThreads], s_j[kThreads];
  const SegItem it = items ? items[blockIdx.x] : inl.it[blockIdx.x];
  const double* P = scratch + (poff ? poff[it.run] : (int64_t)inl.off[it.run]);
  double bs = -1.0; int bL = 0x7fffffff, bj = 0x7fffffff;
  // The run's prefix sums in LDS when they fit (runs are a few thousand bins at most), four lengths per round so that the
  // four window sums, divisions and comparisons of a thread overlap: from global memory, one (L, j) pair at a time, the
  // loop waited for its two loads in every iteration.
  __shared__ double s_P[kBestLds];
  const bool staged = it.len + 1 <= kBestLds;
  if (staged) for (int e = threadIdx.x; e <= it.len; e += kThreads) s_P[e] = P[e];
  __syncthreads();
  const double* Q = staged ? s_P : P;
  for (int L0 = it.Lbeg; L0 < it.Lend; L0 += 4) {
    double dL[4], sq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { dL[u] = (double)(L0 + u); sq[u] = sqrt(dL[u]); }
    for (int j = threadIdx.x; j + L0 <= it.len; j += kThreads) {
      const double pj = Q[j];
      double sum[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int L = L0 + u; sum[u] = (L < it.Lend && j + L <= it.len) ? Q[j + L] - pj : 0.0; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int L = L0 + u;
        if (L < it.Lend && j + L <= it.len) {
          const double score = fabs(sum[u] / dL[u] - tmedian) * sq[u];      // rsi.cpp:1084
          if (seg_better(score, L, j, bs, bL, bj)) { bs = score; bL = L; bj = j; }
        }
      }
    }
  }
  s_s[threadIdx.x] = bs; s_L[threadIdx.x] = bL; s_j[threadIdx.x] = bj;
  __syncthreads();
  for (int d = kThreads / 2; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) {
      if (seg_better(s_s[threadIdx.x + d], s_L[threadIdx.x + d], s_j[threadIdx.x + d], s_s[threadIdx.x], s_L[threadIdx.x], s_j[threadIdx.x])) {
        s_s[threadIdx.x] = s_s[threadIdx.x + d]; s_L[threadIdx.x] = s_L[threadIdx.x + d]; s_j[threadIdx.x] = s_j[threadIdx.x + d];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[blockIdx.x].score = s_s[0]; out[blockIdx.x].start = s_j[0]; out[blockIdx.x].len = s_L[0]; }
}

}  // namespace

static int bounded_grid(int64_t n, int per_block, int cap) { const int g = grid_for(n, per_block); return g > cap ? cap : g; }
void launch_nb_raw(const int64_t* binsum, int64_t nb, int m, int64_t ncompact, double r, float* raw, uint32_t* rawmin_bits,
                   hipStream_t stream) {
  int grid = grid_for(nb, kThreads * 4);
  if (grid > 1024) grid = 1024;
  RSI_LAUNCH(k_nb_raw, dim3(grid), dim3(kThreads), 0, stream, binsum, nb, m, ncompact, r, raw, rawmin_bits);
}
void launch_nb_scale_minmax(float* x, int64_t nb, const uint32_t* rawmin_bits, double med_raw, double del_raw, double dup_raw,
                            double RDmedian, const GridChain& c, GridMedian* out, hipStream_t stream) {
  NbLevels lv{med_raw, del_raw, dup_raw, RDmedian};
  // a bounded grid: every workgroup ends in three atomics on the same words (min, max, arrival); 2400 workgroups for the bins of a
  // 250 Mb chromosome spent 170 of the kernel's 180 us queueing there (9 us for the 590 of a 60 Mb one)
  RSI_LAUNCH(k_nb_scale_mm, dim3(bounded_grid(nb, kThreads * 4, 512)), dim3(kThreads), 0, stream, x, nb, rawmin_bits, lv, c.mm, c.counters, c.cap,
                     c.hist, out);
}
void launch_i32_to_f32_minmax(const int32_t* in, float* out_f, int64_t nb, double center, const GridChain& c, GridMedian* out,
                              hipStream_t stream) {
  RSI_LAUNCH(k_i32_to_f32_mm, dim3(bounded_grid(nb, kThreads * 4, 512)), dim3(kThreads), 0, stream, in, out_f, nb, center, c.mm, c.counters, c.cap,
                     c.hist, out);
}
void launch_minmax_f32(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, MinMaxF* mm,
                       hipStream_t stream) {
  RSI_LAUNCH(k_minmax_f32, dim3(grid_for(nb, kThreads * 16)), dim3(kThreads), 0, stream, x, mask, nb, use_abs, center, mm);
}
void launch_hist_f32(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, double ymin, uint32_t* hist,
                     uint32_t np, hipStream_t stream) {
  const int use_lds = (int)kLdsBins;   // the first kLdsBins buckets in LDS, the rest through global atomics
  const size_t lds = (size_t)(np < kLdsBins ? np : kLdsBins) * 4;
  RSI_ALLOW_FULL_LDS(k_hist_f32);
  // few, long-lived workgroups: every one flushes np counters at the end
  int grid = grid_for(nb, kThreads * kHistRun * 8);
  if (grid > 128) grid = 128;
  RSI_LAUNCH(k_hist_f32, dim3(grid), dim3(kThreads), lds, stream, x, mask, nb, use_abs, center, ymin, hist, np, use_lds);
}
void launch_minmax_plan(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, const double* d_center,
                        const GridChain& c, GridMedian* out, hipStream_t stream) {
  RSI_LAUNCH(k_minmax_plan, dim3(bounded_grid(nb, kThreads * 16, 256)), dim3(kThreads), 0, stream, x, mask, nb, use_abs, center, d_center, c.mm,
                     c.counters, c.cap, c.hist, out);
}
void launch_hist_walk(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, const double* d_center,
                      const GridChain& c, GridMedian* out, const GridExport* ex, const FillList* fill, hipStream_t stream) {
  const uint32_t bins = kLdsBins;
  int grid = grid_for(nb, kThreads * kHistRun * 2);
  if (grid > 256) grid = 256;
  // 16-bit LDS counters while a workgroup's share of the bins cannot make one wrap
  const bool pack16 = (nb + grid - 1) / grid + kHistRun < 65536;
  const size_t lds = pack16 ? (size_t)bins * 2 : (size_t)bins * 4;
  ExportPair e{};
  if (ex) for (int k = 0; k < 2; ++k) { e.src[k] = ex->src[k]; e.dst[k] = ex->dst[k]; e.bytes[k] = (unsigned int)ex->bytes[k]; }
  FillList f{};
  if (fill) f = *fill;
  if (pack16) RSI_LAUNCH(k_hist_walk<true>, dim3(grid), dim3(kThreads), lds, stream, x, mask, nb, use_abs, center, d_center, out, c.hist, c.counters + 1, e, f, bins);
  else { RSI_ALLOW_FULL_LDS(k_hist_walk<false>); RSI_LAUNCH(k_hist_walk<false>, dim3(grid), dim3(kThreads), lds, stream, x, mask, nb, use_abs, center, d_center, out, c.hist, c.counters + 1, e, f, bins); }
}
constexpr int kGTileGrid = 256;   // workgroups of the device-memory-tile form (one workspace slice each) ...
constexpr size_t kGTileBudget = size_t(8) << 30;   // ... fewer where 256 slices would take more than this (Lmax beyond 300 000)
struct ScanShape { int count, kcap, grid; bool gtile, wide; size_t tile_bytes; };
static ScanShape scan_tile_shape(int Lmax) {
  ScanShape sh;
  const int halo = Lmax / 2 + 1;
  sh.count = kScanTile + 2 * halo;
  // block-level cap: floor(log2(Lmax)), at most 6
  sh.kcap = 0;
  while ((2 << sh.kcap) <= Lmax && sh.kcap < 6) ++sh.kcap;
  sh.gtile = Lmax > kScanLdsL;
  sh.wide = Lmax > kScanNarrowL;
  // in LDS: lowered until the tile fits
  if (!sh.gtile) while (sh.kcap > 0 && scan_lds_bytes(sh.count, sh.kcap) + 4 * kThreads * sizeof(int) + 1024 > 160 * 1024) --sh.kcap;
  sh.tile_bytes = scan_lds_bytes(sh.count, sh.kcap, sh.wide ? 4 : 2);
  sh.grid = kGTileGrid;
  if (sh.gtile) {
    const size_t slice = (sh.tile_bytes + 255) & ~size_t(255);
    if (slice * (size_t)sh.grid > kGTileBudget) sh.grid = (int)std::max<size_t>(1, kGTileBudget / slice);
  }
  return sh;
}
size_t scan_tile_workspace_bytes(int Lmax) {
  const ScanShape sh = scan_tile_shape(Lmax);
  return sh.gtile ? (size_t)sh.grid * ((sh.tile_bytes + 255) & ~size_t(255)) : 0;
}
void launch_rsi_scan(const float* T, const int32_t* medint, const ScanParams& sp_in, const double* thr_del, const double* thr_dup,
                     const ScanThr* inl, uint32_t* first_del, uint32_t* first_dup, uint32_t* counters, uint32_t* tiles, void* tile_ws,
                     hipStream_t stream) {
  ScanParams sp = sp_in;
  const ScanShape sh = scan_tile_shape(sp.Lmax);
  const bool gtile = sh.gtile;
  sp.kcap = (int16_t)sh.kcap;
  const size_t tile_bytes = sh.tile_bytes;
  const size_t lds = gtile ? 0 : tile_bytes;
  const unsigned long long gbytes = (tile_bytes + 255) & ~size_t(255);
  const int64_t ntiles = (sp.nb + kScanTile - 1) / kScanTile;
  static const ScanThr none{};
  // RSI_HOT_SCAN_DETECT=0: no detection pass, every tile through the exact sweep in one piece (A/B runs; same marks)
  const char* det_env = getenv("RSI_HOT_SCAN_DETECT");
  // the detection pass stages its stretch and the float thresholds in LDS: up to Lmax = 13 000 or so; longer scans send every
  // tile through the exact sweep
  const size_t dlds = detect_lds_bytes(sp.Lmax);
  const bool detect = tiles != nullptr && !(det_env && atoi(det_env) == 0) && dlds + kThreads * 8 + 64 <= 160 * 1024;
  int parts = 1;
  if (detect || gtile) {
    // Lengths of a listed tile over several workgroups: whole groups of kScanPad lengths, up to kScanPartsMax shares
    const int groups = (sp.Lmax + kScanPad - 1) / kScanPad;
    parts = groups < kScanPartsMax ? groups : kScanPartsMax;
    if (const char* pe = getenv("RSI_HOT_SCAN_PARTS")) { const int v = atoi(pe); if (v >= 1 && v <= groups) parts = v; }
  }
  if (detect) {
    const int dgrid = (int)((sp.nb + kDetBins - 1) / kDetBins);
    if (inl) { RSI_ALLOW_FULL_LDS(k_scan_detect<true>); RSI_LAUNCH(k_scan_detect<true>, dim3(dgrid), dim3(kThreads), dlds, stream, T, sp.nb, sp.Lmax, nullptr, nullptr, tiles, counters, *inl); }
    else { RSI_ALLOW_FULL_LDS(k_scan_detect<false>); RSI_LAUNCH(k_scan_detect<false>, dim3(dgrid), dim3(kThreads), dlds, stream, T, sp.nb, sp.Lmax, thr_del, thr_dup, tiles, counters, none); }
  }
  // workgroups pull (tile, share) tasks from a counter: as many as can be resident, never more than there can be tasks
  int64_t grid = ntiles * parts;
  const int64_t resident = gtile ? sh.grid : 256 * (int64_t)(lds > 80 * 1024 ? 1 : lds > 52 * 1024 ? 2 : lds > 39 * 1024 ? 3 : 4);
  if ((detect || gtile) && grid > resident) grid = resident;
  const uint32_t* list = detect ? tiles : nullptr;
  unsigned char* gws = static_cast<unsigned char*>(tile_ws);
#define RSI_SCAN(INL_, GT_, WIDE_, ...) do { RSI_ALLOW_FULL_LDS((k_rsi_scan<INL_, GT_, WIDE_>));                                             \
    RSI_LAUNCH((k_rsi_scan<INL_, GT_, WIDE_>), dim3((unsigned)grid), dim3(kThreads), lds, stream, T, medint, sp, __VA_ARGS__, first_del, first_dup, \
               counters, list, parts, inl ? *inl : none, gws, gbytes); } while (0)
  if (gtile && sh.wide) RSI_SCAN(false, true, true, thr_del, thr_dup);   // (thresholds ride in the arguments up to Lmax = 223 only)
  else if (gtile) { if (inl) RSI_SCAN(true, true, false, nullptr, nullptr); else RSI_SCAN(false, true, false, thr_del, thr_dup); }
  else { if (inl) RSI_SCAN(true, false, false, nullptr, nullptr); else RSI_SCAN(false, false, false, thr_del, thr_dup); }
#undef RSI_SCAN
}
void launch_level_stop(const uint32_t* first_del, const uint32_t* first_dup, int64_t nb, int32_t Lmax, uint32_t* work, void* both,
                       unsigned int* counter, void* host_copy, size_t host_bytes, hipStream_t stream) {
  const size_t lds = (size_t)(Lmax + 1) * 8;   // 80 KB at -m 1
  if (lds + 256 <= 160 * 1024) {
    RSI_ALLOW_FULL_LDS(k_level_stop<false>);
    RSI_LAUNCH(k_level_stop<false>, dim3(grid_for(nb, kThreads * 16)), dim3(kThreads), lds, stream, first_del, first_dup, nb,
                       Lmax, work, static_cast<uint2*>(both), counter, host_copy, (unsigned int)host_bytes);
  } else {
    RSI_LAUNCH(k_level_stop<true>, dim3(grid_for(nb, kThreads * 16)), dim3(kThreads), 0, stream, first_del, first_dup, nb,
                       Lmax, work, static_cast<uint2*>(both), counter, host_copy, (unsigned int)host_bytes);
  }
}
void launch_resolve_runs(const uint32_t* first_del, const uint32_t* first_dup, const uint32_t* levels, int64_t nb, int32_t* status,
                         int32_t* copy, uint64_t* runs, uint32_t* count, uint32_t cap, unsigned int* counter, void* host_copy,
                         uint32_t host_entries, hipStream_t stream) {
  RSI_LAUNCH(k_resolve_runs, dim3(grid_for(nb, kThreads * 4)), dim3(kThreads), 0, stream, first_del, first_dup, levels, nb, status, copy,
                     runs, count, cap, counter, host_copy, host_entries);
}
void launch_trim_runs(const float* T, int32_t* status, const int32_t* run_start, const int32_t* run_end, const RunsInline* inl, int nruns,
                      double delthr, double addthr, hipStream_t stream) {
  if (nruns <= 0) return;
  static const RunsInline none{};
  RSI_LAUNCH(k_trim_runs, dim3(nruns), dim3(kThreads), 0, stream, T, status, run_start, run_end,
                     nruns, delthr, addthr, inl ? *inl : none);
}

// best_subsegment is driven from the host side in two launches (see pipeline.hip)
void launch_run_prefix(const float* T, const int32_t* run_start, const int32_t* run_end, const RunsInline* inl, int nruns, const int64_t* poff,
                       double* scratch, const int32_t* status, int32_t* status_out, hipStream_t stream) {
  if (nruns <= 0) return;
  static const RunsInline none{};
  RSI_LAUNCH(k_run_prefix, dim3(nruns), dim3(kThreads), 0, stream, T, run_start, run_end, poff, scratch, inl ? *inl : none, status, status_out);
}
void launch_best_items(const void* items, const ItemsInline* inl, int nitems, const int64_t* poff, const double* scratch, double tmedian,
                       BestSeg* out, hipStream_t stream) {
  if (nitems <= 0) return;
  static const ItemsInline none{};
  RSI_LAUNCH(k_best_subsegment, dim3(nitems), dim3(kThreads), 0, stream, reinterpret_cast<const SegItem*>(items), poff,
                     scratch, tmedian, out, inl ? *inl : none);
}

}  // namespace rsik
