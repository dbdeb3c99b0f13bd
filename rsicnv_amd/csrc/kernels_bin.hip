// kernels_bin.hip -- bin-level kernels (gfx950): NB transform, 0.01-grid quantile histograms,
// the RSI scan with its LDS prefix tile, status resolution, marked runs and the max-score
// sub-segment search.  These work on nb = n'/m bins (about 1 % of the bases), so they are
// latency/ALU-bound rather than HBM-bound; DESIGN.md section 4 gives each one's budget.
//
// Built with -ffp-contract=off and IEEE division/sqrt: every double expression below has to round
// like the reference's x86-64 build.
#include "kernels.h"

namespace rsik {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxGrid = 256 * 8;

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
  if (lane_id() == 0 && kmin != 0xffffffffu) atomicMin(rawmin_key, kmin);
}

// rsi.cpp:1176-1185: subtract the minimum, rescale to the depth scale, overwrite bins 0..2
__global__ __launch_bounds__(kThreads) void k_nb_scale(float* __restrict__ x, int64_t nb, double tmin, double med_nbt,
                                                       double med, float lev0, float lev1, float lev2) {
  for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < nb; b += (int64_t)gridDim.x * kThreads) {
    float v = x[b];
    v = (float)((double)v - tmin);
    v = (float)((double)v / med_nbt * med);
    if (b == 0) v = lev0;
    if (b == 1) v = lev1;
    if (b == 2) v = lev2;
    x[b] = v;
  }
}

__global__ __launch_bounds__(kThreads) void k_i32_to_f32(const int32_t* __restrict__ in, float* __restrict__ out, int64_t nb) {
  for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < nb; b += (int64_t)gridDim.x * kThreads)
    out[b] = (float)in[b];
}

// ------------------------------------------------------------------------------------------
// K6  quantile histograms of float arrays on the 0.01 grid (partition_stat_tp, wufunctions.cpp:364-424)
__device__ inline float sel_value(const float* __restrict__ x, int64_t i, int use_abs, double center) {
  const float v = x[i];
  return use_abs ? (float)fabs((double)v - center) : v;   // RDtmp[i] = abs(RDtrans[i]-tmedian), rsi.cpp:1276
}

__global__ __launch_bounds__(kThreads) void k_minmax_f32(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                         int64_t nb, int use_abs, double center, MinMaxF* __restrict__ mm) {
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
  if (lane_id() == 0) {
    if (kmin != 0xffffffffu) atomicMin(&mm->min_bits, kmin);
    if (kmax != 0) atomicMax(&mm->max_bits, kmax);
    if (bad) atomicOr(&mm->nonfinite, 1u);
  }
}

constexpr uint32_t kLdsBins = 12288;   // 48 KB of LDS counters

__global__ __launch_bounds__(kThreads) void k_hist_f32(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                       int64_t nb, int use_abs, double center, double ymin,
                                                       uint32_t* __restrict__ hist, uint32_t np, int use_lds) {
  extern __shared__ unsigned int s_h[];
  if (use_lds) { for (uint32_t e = threadIdx.x; e < np; e += kThreads) s_h[e] = 0; __syncthreads(); }
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nb; i += (int64_t)gridDim.x * kThreads) {
    if (mask && mask[i] != 0) continue;
    const float v = sel_value(x, i, use_abs, center);
    const double idx = ((double)v - ymin) / 0.01 + 0.5;      // wufunctions.cpp:396
    uint32_t k = (uint32_t)(unsigned long long)idx;
    if (k >= np) k = np - 1;                                 // cannot happen (np = range/dy + 2)
    if (use_lds) atomicAdd(&s_h[k], 1u); else atomicAdd(&hist[k], 1u);
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < np; e += kThreads) { const unsigned int c = s_h[e]; if (c) atomicAdd(&hist[e], c); }
  }
}

// ------------------------------------------------------------------------------------------
// K7  RSI scan.  One workgroup per tile of kScanTile bins.  The tile plus a halo of Lmax/2+1 bins
// each side is reduced to an exact double prefix in LDS, so the window sum for any (bin, L) is
// one subtraction; the per-L score test is folded on the host into a threshold on that sum
// (hit iff sum <= thr_del[L] / sum >= thr_dup[L]).  Hits are rare and handled by the whole wave:
// exact window median of the bin medians, the four trim walks, then atomicMin(L) on the marked
// bins ("smallest L wins", App. A Q14).
constexpr int kScanTile = 1024;

__device__ inline uint32_t ld_relaxed(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // served from L2, never a stale L1 line
}

// Exact median of medint[w0 .. w0+L-1] (alglib samplemedian semantics), computed by the wave.
__device__ inline double wave_window_median(const int32_t* __restrict__ medint, int64_t w0, int L) {
  const int lane = lane_id();
  int lo = 0x7fffffff, hi = (int)0x80000000;
  for (int j = lane; j < L; j += 64) { const int v = medint[w0 + j]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  for (int d = 32; d >= 1; d >>= 1) {
    const int a = __shfl_xor(lo, d), b = __shfl_xor(hi, d);
    lo = a < lo ? a : lo; hi = b > hi ? b : hi;
  }
  const int klo = (L - 1) / 2 + 1;   // rank (1-based) of the lower middle element
  while (lo < hi) {
    const int mid = (int)(((long long)lo + (long long)hi) >> 1);
    int c = 0;
    for (int j = lane; j < L; j += 64) c += medint[w0 + j] <= mid;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (c >= klo) hi = mid; else lo = mid + 1;
  }
  const int a = lo;
  if (L & 1) return (double)a;
  // even L: mean of the two middle order statistics
  int c = 0, nxt = 0x7fffffff;
  for (int j = lane; j < L; j += 64) { const int v = medint[w0 + j]; c += v <= a; if (v > a && v < nxt) nxt = v; }
  for (int d = 32; d >= 1; d >>= 1) { c += __shfl_xor(c, d); const int o = __shfl_xor(nxt, d); nxt = o < nxt ? o : nxt; }
  const int b = (c >= klo + 1) ? a : nxt;
  return 0.5 * ((double)a + (double)b);
}

// Walk from `pos` in direction dir (+1/-1) while the predicate holds; returns the first position
// where it fails, or -1 / nb when the walk leaves the array (the reference would abort there).
template <class Pred>
__device__ inline int64_t wave_walk(int64_t pos, int dir, int64_t nb, Pred pred) {
  const int lane = lane_id();
  while (true) {
    const int64_t idx = pos + (int64_t)dir * lane;
    const bool inside = idx >= 0 && idx < nb;
    const bool go = inside && pred(idx);
    const unsigned long long stop = __ballot(!go);
    if (stop) {
      const int first = __ffsll((long long)stop) - 1;
      return pos + (int64_t)dir * first;
    }
    pos += (int64_t)dir * 64;
  }
}

__device__ inline void wave_process_hit(const float* __restrict__ T, const int32_t* __restrict__ medint,
                                        const ScanParams& sp, int64_t bi, int L, bool is_del,
                                        uint32_t* __restrict__ first, uint32_t* __restrict__ counters) {
  const int lane = lane_id();
  const int64_t w0 = bi - L / 2;
  // nothing to do when every bin of the window already carries a mark with a length <= L
  {
    uint32_t worst = 0;
    for (int j = lane; j < L; j += 64) { const uint32_t f = ld_relaxed(first + w0 + j); worst = f > worst ? f : worst; }
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(worst, d); worst = o > worst ? o : worst; }
    if (worst <= (uint32_t)L) return;
  }
  const double med = wave_window_median(medint, w0, L);
  const double lim = is_del ? sp.lim_del : sp.lim_dup;
  if (is_del ? (med > lim) : (med < lim)) return;                 // rsi.cpp:1206 / 1236
  const double tmed = sp.tmedian;
  int64_t i1 = w0, i2 = w0 + L - 1;
  if (is_del) {                                                    // rsi.cpp:1211-1214
    i1 = wave_walk(i1, +1, sp.nb, [&](int64_t q) { return (double)T[q] > tmed; });
    if (i1 >= 0 && i1 < sp.nb) i1 = wave_walk(i1, +1, sp.nb, [&](int64_t q) { return (double)medint[q] > lim; });
    i2 = wave_walk(i2, -1, sp.nb, [&](int64_t q) { return (double)T[q] > tmed; });
    if (i2 >= 0 && i2 < sp.nb) i2 = wave_walk(i2, -1, sp.nb, [&](int64_t q) { return (double)medint[q] > lim; });
  } else {                                                         // rsi.cpp:1241-1244
    i1 = wave_walk(i1, +1, sp.nb, [&](int64_t q) { return (double)T[q] < tmed; });
    if (i1 >= 0 && i1 < sp.nb) i1 = wave_walk(i1, +1, sp.nb, [&](int64_t q) { return (double)medint[q] < lim; });
    i2 = wave_walk(i2, -1, sp.nb, [&](int64_t q) { return (double)T[q] < tmed; });
    if (i2 >= 0 && i2 < sp.nb) i2 = wave_walk(i2, -1, sp.nb, [&](int64_t q) { return (double)medint[q] < lim; });
  }
  if (i1 < 0 || i1 >= sp.nb || i2 < 0 || i2 >= sp.nb) {            // App. A Q12: mark nothing, count it
    if (lane == 0) atomicAdd(&counters[0], 1u);
    return;
  }
  for (int64_t j = i1 + lane; j <= i2; j += 64) atomicMin(&first[j], (uint32_t)L);
}

__global__ __launch_bounds__(kThreads) void k_rsi_scan(const float* __restrict__ T, const int32_t* __restrict__ medint,
                                                       ScanParams sp, const double* __restrict__ thr_del,
                                                       const double* __restrict__ thr_dup,
                                                       uint32_t* __restrict__ first_del, uint32_t* __restrict__ first_dup,
                                                       uint32_t* __restrict__ counters) {
  extern __shared__ __align__(16) double sm[];
  const int Lmax = sp.Lmax;
  const int halo = Lmax / 2 + 1;
  const int count = kScanTile + 2 * halo;          // staged bins
  double* P = sm;                                  // count + 1 prefix entries
  double* s_del = sm + count + 1;                  // Lmax + 1
  double* s_dup = s_del + Lmax + 1;                // Lmax + 1
  double* s_tot = s_dup + Lmax + 1;                // kThreads chunk totals
  const int64_t tile_start = (int64_t)blockIdx.x * kScanTile;
  const int64_t lo = tile_start - halo;
  for (int e = threadIdx.x; e <= Lmax; e += kThreads) { s_del[e] = thr_del[e]; s_dup[e] = thr_dup[e]; }
  // ---- exact prefix over the staged bins: serial chunk per thread, then a scan of the chunk totals ----
  const int chunk = (count + kThreads - 1) / kThreads;
  const int c0 = threadIdx.x * chunk;
  double run = 0.0;
  unsigned int inexact = 0;
  for (int e = c0; e < c0 + chunk && e < count; ++e) {
    const int64_t g = lo + e;
    const float v = (g >= 0 && g < sp.nb) ? T[g] : 0.0f;
    // exact-sum precondition: 0, or 2^-10 <= |v| < 2^20; counted once per owning tile
    const float av = fabsf(v);
    if (!(av == 0.0f || (av >= 0.0009765625f && av < 1048576.0f)) && g >= tile_start && g < tile_start + kScanTile) inexact++;
    run += (double)v;
    P[e + 1] = run;
  }
  s_tot[threadIdx.x] = run;
  if (threadIdx.x == 0) P[0] = 0.0;
  __syncthreads();
  if (threadIdx.x < 64) {   // wave 0 turns the 256 totals into exclusive offsets
    double carry = 0.0;
    for (int k = 0; k < kThreads / 64; ++k) {
      const double mine = s_tot[k * 64 + threadIdx.x];
      double incl = mine;
      for (int d = 1; d < 64; d <<= 1) { const double up = __shfl_up(incl, d); if ((int)threadIdx.x >= d) incl += up; }
      s_tot[k * 64 + threadIdx.x] = carry + incl - mine;
      carry += __shfl(incl, 63);
    }
  }
  __syncthreads();
  {
    const double off = s_tot[threadIdx.x];
    for (int e = c0; e < c0 + chunk && e < count; ++e) P[e + 1] += off;
  }
  for (int d = 32; d >= 1; d >>= 1) inexact += __shfl_xor(inexact, d);
  if (lane_id() == 0 && inexact) atomicAdd(&counters[1], inexact);
  __syncthreads();

  // ---- evaluate every (bin, L) of the tile ----
  for (int r = 0; r < kScanTile / kThreads; ++r) {
    const int64_t i = tile_start + r * kThreads + threadIdx.x;
    const int rel = (int)(i - lo);   // index of bin i among the staged bins
    for (int L = 1; L <= Lmax; ++L) {
      const int h = L / 2;
      // the reference visits i in [L/2+1, nb-L/2-2] (rsi.cpp:1204)
      const bool visit = i < sp.nb && i >= h + 1 && i < sp.nb - h - 1;
      bool hd = false, hu = false;
      if (visit) {
        const double sum = P[rel - h + L] - P[rel - h];
        hd = sum <= s_del[L];
        hu = sum >= s_dup[L];
      }
      unsigned long long any = __ballot(hd || hu);
      while (any) {
        const int src = __ffsll((long long)any) - 1;
        any &= any - 1;
        const int64_t bi = __shfl(i, src);
        const int isdel = __shfl((int)hd, src);
        wave_process_hit(T, medint, sp, bi, L, isdel != 0, isdel ? first_del : first_dup, counters);
      }
    }
  }
}

// histogram over L of first[] (bins whose exclude[] <= exclude_max are skipped: DEL marks win)
__global__ __launch_bounds__(kThreads) void k_level_hist(const uint32_t* __restrict__ first,
                                                         const uint32_t* __restrict__ exclude, uint32_t exclude_max,
                                                         int64_t nb, int32_t Lmax, uint32_t* __restrict__ hist) {
  extern __shared__ unsigned int s_l[];
  for (int e = threadIdx.x; e <= Lmax; e += kThreads) s_l[e] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nb; i += (int64_t)gridDim.x * kThreads) {
    const uint32_t f = first[i];
    if (f > (uint32_t)Lmax) continue;
    if (exclude && exclude[i] <= exclude_max) continue;
    atomicAdd(&s_l[f], 1u);
  }
  __syncthreads();
  for (int e = threadIdx.x; e <= Lmax; e += kThreads) { const unsigned int c = s_l[e]; if (c) atomicAdd(&hist[e], c); }
}

__global__ __launch_bounds__(kThreads) void k_resolve_status(const uint32_t* __restrict__ first_del,
                                                             const uint32_t* __restrict__ first_dup, uint32_t ldel,
                                                             uint32_t ldup, int64_t nb, int32_t* __restrict__ status) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nb; i += (int64_t)gridDim.x * kThreads) {
    const uint32_t fd = first_del[i], fu = first_dup[i];
    int s = 0;
    if (fd <= ldel) s = -(int)fd;
    else if (fu <= ldup) s = (int)fu;
    status[i] = s;
  }
}

// ------------------------------------------------------------------------------------------
// K9  marked runs (get_continuous_segments with d = 1, rsi.cpp:291-327): boundaries only
__global__ __launch_bounds__(kThreads) void k_find_runs(const int32_t* __restrict__ status, int64_t nb,
                                                        uint64_t* __restrict__ runs, uint32_t* __restrict__ count,
                                                        uint32_t cap) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nb; i += (int64_t)gridDim.x * kThreads) {
    const int s = status[i];
    if (s == 0) continue;
    const int p = i > 0 ? status[i - 1] : 0, q = i + 1 < nb ? status[i + 1] : 0;
    const bool is_start = p == 0 || ((p > 0) != (s > 0));
    const bool is_end = q == 0 || ((q > 0) != (s > 0));
    if (is_start) { const uint32_t k = atomicAdd(count, 1u); if (k < cap) runs[k] = ((uint64_t)i << 1); }
    if (is_end) { const uint32_t k = atomicAdd(count, 1u); if (k < cap) runs[k] = ((uint64_t)i << 1) | 1u; }
  }
}

// edge trimming of filterstatus (rsi.cpp:1023-1044): one thread per run, runs are disjoint
__global__ __launch_bounds__(kThreads) void k_trim_runs(const float* __restrict__ T, int32_t* __restrict__ status,
                                                        const int32_t* __restrict__ run_start,
                                                        const int32_t* __restrict__ run_end, int nruns, double delthr,
                                                        double addthr) {
  const int r = blockIdx.x * kThreads + threadIdx.x;
  if (r >= nruns) return;
  int i1 = run_start[r], i2 = run_end[r];
  while (((double)T[i1] > delthr && status[i1] < 0) || ((double)T[i1] < addthr && status[i1] > 0)) {
    status[i1] = 0; ++i1; if (i1 >= i2) break;
  }
  while (((double)T[i2] > delthr && status[i2] < 0) || ((double)T[i2] < addthr && status[i2] > 0)) {
    status[i2] = 0; --i2; if (i2 <= i1) break;
  }
}

// ------------------------------------------------------------------------------------------
// K10  max-score sub-segment of each run (get_rsi_segments, rsi.cpp:1060-1117).
// Step 1: exact double prefix of the run's values into scratch (one workgroup per run).
__global__ __launch_bounds__(kThreads) void k_run_prefix(const float* __restrict__ T, const int32_t* __restrict__ run_start,
                                                         const int32_t* __restrict__ run_end,
                                                         const int64_t* __restrict__ poff, double* __restrict__ scratch) {
  __shared__ double s_tot[kThreads];
  const int r = blockIdx.x;
  const int s = run_start[r], len = run_end[r] - run_start[r] + 1;
  double* P = scratch + poff[r];
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
      double incl = mine;
      for (int d = 1; d < 64; d <<= 1) { const double up = __shfl_up(incl, d); if ((int)threadIdx.x >= d) incl += up; }
      s_tot[k * 64 + threadIdx.x] = carry + incl - mine;
      carry += __shfl(incl, 63);
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

__global__ __launch_bounds__(kThreads) void k_best_subsegment(const SegItem* __restrict__ items,
                                                              const int64_t* __restrict__ poff,
                                                              const double* __restrict__ scratch, double tmedian,
                                                              BestSeg* __restrict__ out) {
  __shared__ double s_s[kThreads];
  __shared__ int s_L[kThreads], s_j[kThreads];
  const SegItem it = items[blockIdx.x];
  const double* P = scratch + poff[it.run];
  double bs = -1.0; int bL = 0x7fffffff, bj = 0x7fffffff;
  for (int L = it.Lbeg; L < it.Lend; ++L) {
    const double dL = (double)L, sq = sqrt(dL);
    for (int j = threadIdx.x; j + L <= it.len; j += kThreads) {
      const double sum = P[j + L] - P[j];
      const double score = fabs(sum / dL - tmedian) * sq;      // rsi.cpp:1084
      if (seg_better(score, L, j, bs, bL, bj)) { bs = score; bL = L; bj = j; }
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

void launch_nb_raw(const int64_t* binsum, int64_t nb, int m, int64_t ncompact, double r, float* raw, uint32_t* rawmin_bits,
                   hipStream_t stream) {
  hipLaunchKernelGGL(k_nb_raw, dim3(grid_for(nb, kThreads)), dim3(kThreads), 0, stream, binsum, nb, m, ncompact, r, raw, rawmin_bits);
}
void launch_nb_scale(float* x, int64_t nb, double tmin, double med_nbt, double med, float lev0, float lev1, float lev2,
                     hipStream_t stream) {
  hipLaunchKernelGGL(k_nb_scale, dim3(grid_for(nb, kThreads)), dim3(kThreads), 0, stream, x, nb, tmin, med_nbt, med, lev0, lev1, lev2);
}
void launch_i32_to_f32(const int32_t* in, float* out, int64_t nb, hipStream_t stream) {
  hipLaunchKernelGGL(k_i32_to_f32, dim3(grid_for(nb, kThreads)), dim3(kThreads), 0, stream, in, out, nb);
}
void launch_minmax_f32(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, MinMaxF* mm,
                       hipStream_t stream) {
  hipLaunchKernelGGL(k_minmax_f32, dim3(grid_for(nb, kThreads * 4)), dim3(kThreads), 0, stream, x, mask, nb, use_abs, center, mm);
}
void launch_hist_f32(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, double ymin, uint32_t* hist,
                     uint32_t np, hipStream_t stream) {
  const int use_lds = np <= kLdsBins;
  hipLaunchKernelGGL(k_hist_f32, dim3(grid_for(nb, kThreads * 16)), dim3(kThreads), use_lds ? (size_t)np * 4 : 0, stream, x, mask,
                     nb, use_abs, center, ymin, hist, np, use_lds);
}
void launch_rsi_scan(const float* T, const int32_t* medint, const ScanParams& sp, const double* thr_del, const double* thr_dup,
                     uint32_t* first_del, uint32_t* first_dup, uint32_t* counters, hipStream_t stream) {
  const int halo = sp.Lmax / 2 + 1;
  const size_t lds = ((size_t)(kScanTile + 2 * halo + 1) + 2 * (size_t)(sp.Lmax + 1) + kThreads) * sizeof(double);
  const int grid = (int)((sp.nb + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(k_rsi_scan, dim3(grid), dim3(kThreads), lds, stream, T, medint, sp, thr_del, thr_dup, first_del, first_dup, counters);
}
void launch_level_hist(const uint32_t* first, const uint32_t* exclude, uint32_t exclude_max, int64_t nb, int32_t Lmax,
                       uint32_t* hist, hipStream_t stream) {
  hipLaunchKernelGGL(k_level_hist, dim3(grid_for(nb, kThreads * 16)), dim3(kThreads), (size_t)(Lmax + 1) * 4, stream, first, exclude,
                     exclude_max, nb, Lmax, hist);
}
void launch_resolve_status(const uint32_t* first_del, const uint32_t* first_dup, uint32_t ldel, uint32_t ldup, int64_t nb,
                           int32_t* status, hipStream_t stream) {
  hipLaunchKernelGGL(k_resolve_status, dim3(grid_for(nb, kThreads)), dim3(kThreads), 0, stream, first_del, first_dup, ldel, ldup, nb, status);
}
void launch_find_runs(const int32_t* status, int64_t nb, uint64_t* runs, uint32_t* count, uint32_t cap, hipStream_t stream) {
  hipLaunchKernelGGL(k_find_runs, dim3(grid_for(nb, kThreads)), dim3(kThreads), 0, stream, status, nb, runs, count, cap);
}
void launch_trim_runs(const float* T, int32_t* status, const int32_t* run_start, const int32_t* run_end, int nruns,
                      double delthr, double addthr, hipStream_t stream) {
  if (nruns <= 0) return;
  hipLaunchKernelGGL(k_trim_runs, dim3((nruns + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, T, status, run_start, run_end,
                     nruns, delthr, addthr);
}

// best_subsegment is driven from the host side in two launches (see pipeline.hip)
void launch_run_prefix(const float* T, const int32_t* run_start, const int32_t* run_end, int nruns, const int64_t* poff,
                       double* scratch, hipStream_t stream) {
  if (nruns <= 0) return;
  hipLaunchKernelGGL(k_run_prefix, dim3(nruns), dim3(kThreads), 0, stream, T, run_start, run_end, poff, scratch);
}
void launch_best_items(const void* items, int nitems, const int64_t* poff, const double* scratch, double tmedian, BestSeg* out,
                       hipStream_t stream) {
  if (nitems <= 0) return;
  hipLaunchKernelGGL(k_best_subsegment, dim3(nitems), dim3(kThreads), 0, stream, reinterpret_cast<const SegItem*>(items), poff,
                     scratch, tmedian, out);
}

}  // namespace rsik
