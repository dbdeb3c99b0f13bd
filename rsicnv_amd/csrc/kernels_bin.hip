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
  __shared__ uint32_t s_min[kThreads / 64], s_max[kThreads / 64], s_bad[kThreads / 64];
  if (lane_id() == 0) { s_min[threadIdx.x >> 6] = kmin; s_max[threadIdx.x >> 6] = kmax; s_bad[threadIdx.x >> 6] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {   // one set of atomics per workgroup
    for (int w = 1; w < kThreads / 64; ++w) { kmin = s_min[w] < kmin ? s_min[w] : kmin; kmax = s_max[w] > kmax ? s_max[w] : kmax; bad |= s_bad[w]; }
    if (kmin != 0xffffffffu) atomicMin(&mm->min_bits, kmin);
    if (kmax != 0) atomicMax(&mm->max_bits, kmax);
    if (bad) atomicOr(&mm->nonfinite, 1u);
  }
}

constexpr uint32_t kLdsBins = 32768;   // 128 KB of LDS counters (one workgroup per CU): covers a value range of 327

// Each thread takes kHistRun consecutive elements and merges equal neighbouring buckets before the
// atomic: bin medians (-MED) are small integers, so long runs land in one bucket and a plain
// one-atomic-per-element histogram serialises on it (271 ms per 3 Gb step with global atomics).
constexpr int kHistRun = 8;

__global__ __launch_bounds__(kThreads) void k_hist_f32(const float* __restrict__ x, const int32_t* __restrict__ mask,
                                                       int64_t nb, int use_abs, double center, double ymin,
                                                       uint32_t* __restrict__ hist, uint32_t np, int use_lds) {
  extern __shared__ unsigned int s_h[];
  if (use_lds) { for (uint32_t e = threadIdx.x; e < np; e += kThreads) s_h[e] = 0; __syncthreads(); }
  const int64_t nchunks = (nb + kHistRun - 1) / kHistRun;
  for (int64_t c = (int64_t)blockIdx.x * kThreads + threadIdx.x; c < nchunks; c += (int64_t)gridDim.x * kThreads) {
    uint32_t pend_k = 0xffffffffu, pend_c = 0;
    const int64_t i0 = c * kHistRun, i1 = i0 + kHistRun < nb ? i0 + kHistRun : nb;
    for (int64_t i = i0; i < i1; ++i) {
      if (mask && mask[i] != 0) continue;
      const float v = sel_value(x, i, use_abs, center);
      const double idx = ((double)v - ymin) / 0.01 + 0.5;      // wufunctions.cpp:396
      uint32_t k = (uint32_t)(unsigned long long)idx;
      if (k >= np) k = np - 1;                                 // cannot happen (np = range/dy + 2)
      if (k != pend_k) {
        if (pend_c) { if (use_lds) atomicAdd(&s_h[pend_k], pend_c); else atomicAdd(&hist[pend_k], pend_c); }
        pend_k = k; pend_c = 0;
      }
      ++pend_c;
    }
    if (pend_c) { if (use_lds) atomicAdd(&s_h[pend_k], pend_c); else atomicAdd(&hist[pend_k], pend_c); }
  }
  if (use_lds) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < np; e += kThreads) { const unsigned int c = s_h[e]; if (c) atomicAdd(&hist[e], c); }
  }
}

// ------------------------------------------------------------------------------------------
// K7  RSI scan.  One workgroup per tile of kScanTile bins (one per thread).  The tile plus a halo of Lmax/2+1 bins
// each side is staged in LDS: an exact double prefix of the transformed values (so the window sum
// for any (bin, L) is one subtraction), the values and bin medians themselves for the trim walks,
// and two integer prefixes that turn the exact window-median test into a count difference.  The
// per-L score test is folded on the host into a threshold on the window sum (hit iff
// sum <= thr_del[L] / sum >= thr_dup[L]).  Every lane owns a bin and walks L = 1..Lmax; a hit marks
// its trimmed interval with atomicMin(L) in LDS ("smallest L wins", App. A Q14) and the tile's
// marks are merged into HBM at the end.  Nothing in the hit path leaves LDS.
constexpr int kScanTile = 256;    // one bin per lane: the tile's critical path is one lane's walk over L
constexpr uint32_t kUnmarked = 0xffffffffu;

struct ScanLds {
  double* P;        // count + 1
  double* tdel;     // Lmax + 1
  double* tdup;     // Lmax + 1
  double* tot;      // kThreads
  float* T;         // count
  int* M;           // count
  int* CL;          // count + 1: # staged bins before e with medint <= floor(lim_del)
  int* CG;          // count + 1: # staged bins before e with medint >= ceil(lim_dup)
  uint32_t* FD;     // count
  uint32_t* FU;     // count
};

// One sweep's hit for the lane's window [w0, w0+L-1] (staged indices).  DEL: is_del = true.
// vlo/vhi: staged indices inside the chromosome are [vlo, vhi).  Returns the updated "largest mark
// in the window" (see the skip rule in the caller).
__device__ inline uint32_t lane_hit(const ScanLds& S, int w0, int L, bool is_del, int thr_int, double lim, double tmed,
                                    int vlo, int vhi, bool at_start, bool at_end, uint32_t* F, uint32_t* counters) {
  const int* C = is_del ? S.CL : S.CG;
  const int c = C[w0 + L] - C[w0];
  bool pass;
  if (L & 1) {
    pass = c >= (L + 1) / 2;                       // middle order statistic on the right side of the limit
  } else {
    const int hh = L / 2;
    if (c >= hh + 1) pass = true;
    else if (c <= hh - 1) pass = false;
    else {                                         // the two middle elements straddle the limit: need their values
      int a, b;
      if (is_del) {                                // a = max{x <= thr}, b = min{x > thr}
        a = (int)0x80000000; b = 0x7fffffff;
        for (int j = w0; j < w0 + L; ++j) { const int x = S.M[j]; if (x <= thr_int) a = x > a ? x : a; else b = x < b ? x : b; }
        pass = !(0.5 * ((double)a + (double)b) > lim);          // rsi.cpp:1206
      } else {                                     // a = max{x < thr}, b = min{x >= thr}
        a = (int)0x80000000; b = 0x7fffffff;
        for (int j = w0; j < w0 + L; ++j) { const int x = S.M[j]; if (x >= thr_int) b = x < b ? x : b; else a = x > a ? x : a; }
        pass = !(0.5 * ((double)a + (double)b) < lim);          // rsi.cpp:1236
      }
    }
  }
  if (!pass) return kUnmarked;   // no marks from this hit, and no knowledge of the window's marks
  uint32_t wmax = 0;
  // trim walks in the reference's order (rsi.cpp:1211-1214 / 1241-1244), bounded to the chromosome
  int i1 = w0, i2 = w0 + L - 1;
  if (is_del) {
    while (i1 < vhi && (double)S.T[i1] > tmed) ++i1;
    while (i1 < vhi && S.M[i1] > thr_int) ++i1;
    while (i2 >= vlo && (double)S.T[i2] > tmed) --i2;
    while (i2 >= vlo && S.M[i2] > thr_int) --i2;
  } else {
    while (i1 < vhi && (double)S.T[i1] < tmed) ++i1;
    while (i1 < vhi && S.M[i1] < thr_int) ++i1;
    while (i2 >= vlo && (double)S.T[i2] < tmed) --i2;
    while (i2 >= vlo && S.M[i2] < thr_int) --i2;
  }
  if (i1 >= vhi || i2 < vlo) {
    // left the staged range: the marked interval is empty either way.  Leaving the chromosome
    // itself is where the reference aborts (App. A Q12): count those.
    if ((i1 >= vhi && at_end) || (i2 < vlo && at_start)) atomicAdd(&counters[0], 1u);
    i1 = 1; i2 = 0;
  }
  // mark the trimmed interval; the window's largest mark is only worth knowing when the interval
  // covers the whole window (otherwise the untouched rest keeps the lane from skipping anyway)
  const bool full = i1 == w0 && i2 == w0 + L - 1;
  for (int j = i1; j <= i2; ++j) {
    uint32_t v = F[j];
    if (v > (uint32_t)L) { atomicMin(&F[j], (uint32_t)L); v = (uint32_t)L; }
    wmax = v > wmax ? v : wmax;
  }
  return full ? wmax : kUnmarked;
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
  ScanLds S;
  S.P = sm;
  S.tdel = S.P + count + 1;
  S.tdup = S.tdel + Lmax + 1;
  S.tot = S.tdup + Lmax + 1;
  S.T = reinterpret_cast<float*>(S.tot + kThreads);
  S.M = reinterpret_cast<int*>(S.T + count);
  S.CL = S.M + count;
  S.CG = S.CL + count + 1;
  S.FD = reinterpret_cast<uint32_t*>(S.CG + count + 1);
  S.FU = S.FD + count;
  const int64_t tile_start = (int64_t)blockIdx.x * kScanTile;
  const int64_t lo = tile_start - halo;
  const int vlo = lo < 0 ? (int)(-lo) : 0;
  const int vhi = (lo + count > sp.nb) ? (int)(sp.nb - lo) : count;
  const bool at_start = lo <= 0, at_end = lo + count >= sp.nb;
  // integer forms of the median limits: x > lim_del <=> x > fl_del ; x < lim_dup <=> x < ce_dup
  const int fl_del = (int)floor(sp.lim_del), ce_dup = (int)ceil(sp.lim_dup);
  for (int e = threadIdx.x; e <= Lmax; e += kThreads) { S.tdel[e] = thr_del[e]; S.tdup[e] = thr_dup[e]; }
  // ---- stage + exact prefixes: serial chunk per thread, then a scan of the 256 chunk totals ----
  const int chunk = (count + kThreads - 1) / kThreads;
  const int c0 = threadIdx.x * chunk;
  double run = 0.0;
  int runl = 0, rung = 0;
  unsigned int inexact = 0;
  for (int e = c0; e < c0 + chunk && e < count; ++e) {
    const bool in = e >= vlo && e < vhi;
    const float v = in ? T[lo + e] : 0.0f;
    const int mi = in ? medint[lo + e] : 0;
    // exact-sum precondition: 0, or 2^-10 <= |v| < 2^20; counted once, by the owning tile
    const float av = fabsf(v);
    if (!(av == 0.0f || (av >= 0.0009765625f && av < 1048576.0f)) && e >= halo && e < halo + kScanTile) inexact++;
    S.T[e] = v; S.M[e] = mi; S.FD[e] = kUnmarked; S.FU[e] = kUnmarked;
    run += (double)v;
    runl += (in && mi <= fl_del);
    rung += (in && mi >= ce_dup);
    S.P[e + 1] = run; S.CL[e + 1] = runl; S.CG[e + 1] = rung;
  }
  // pack the three chunk totals for the cross-thread scan (counts are exact in double)
  S.tot[threadIdx.x] = run;
  __shared__ int s_cl[kThreads], s_cg[kThreads];
  s_cl[threadIdx.x] = runl; s_cg[threadIdx.x] = rung;
  if (threadIdx.x == 0) { S.P[0] = 0.0; S.CL[0] = 0; S.CG[0] = 0; }
  __syncthreads();
  if (threadIdx.x < 64) {   // wave 0 turns the 256 totals into exclusive offsets
    double carry = 0.0; int carl = 0, carg = 0;
    for (int k = 0; k < kThreads / 64; ++k) {
      const int idx = k * 64 + threadIdx.x;
      const double mine = S.tot[idx]; const int ml = s_cl[idx], mg = s_cg[idx];
      double incl = mine; int il = ml, ig = mg;
      for (int d = 1; d < 64; d <<= 1) {
        const double up = __shfl_up(incl, d); const int ul = __shfl_up(il, d), ug = __shfl_up(ig, d);
        if ((int)threadIdx.x >= d) { incl += up; il += ul; ig += ug; }
      }
      S.tot[idx] = carry + incl - mine; s_cl[idx] = carl + il - ml; s_cg[idx] = carg + ig - mg;
      carry += __shfl(incl, 63); carl += __shfl(il, 63); carg += __shfl(ig, 63);
    }
  }
  __syncthreads();
  {
    const double off = S.tot[threadIdx.x]; const int ol = s_cl[threadIdx.x], og = s_cg[threadIdx.x];
    for (int e = c0; e < c0 + chunk && e < count; ++e) { S.P[e + 1] += off; S.CL[e + 1] += ol; S.CG[e + 1] += og; }
  }
  for (int d = 32; d >= 1; d >>= 1) inexact += __shfl_xor(inexact, d);
  if (lane_id() == 0 && inexact) atomicAdd(&counters[1], inexact);
  __syncthreads();

  // ---- every (bin, L) of the tile; lanes are independent ----
  const double tmed = sp.tmedian;
  for (int r = 0; r < kScanTile / kThreads; ++r) {
    const int64_t i = tile_start + r * kThreads + threadIdx.x;
    const int rel = (int)(i - lo);   // staged index of bin i
    // skip rule: wd / wu hold the largest mark in the lane's window as of length ld / lu; while the
    // window grows by one bin per L they are extended incrementally, otherwise recomputed by lane_hit.
    // A hit whose whole window already carries marks <= L cannot change anything (the trimmed
    // interval lies inside the window), so it is skipped.
    uint32_t wd = kUnmarked, wu = kUnmarked; int ld = -1, lu = -1;
    for (int L = 1; L <= Lmax; ++L) {
      const int h = L / 2;
      // the reference visits i in [L/2+1, nb-L/2-2] (rsi.cpp:1204)
      if (!(i < sp.nb && i >= h + 1 && i < sp.nb - h - 1)) continue;
      const int w0 = rel - h;
      const double sum = S.P[w0 + L] - S.P[w0];
      const int grown = (L & 1) ? w0 + L - 1 : w0;   // the bin the window gained going from L-1 to L
      if (sum <= S.tdel[L]) {
        if (ld == L - 1) { const uint32_t v = S.FD[grown]; wd = v > wd ? v : wd; } else wd = kUnmarked;
        if (wd > (uint32_t)L) wd = lane_hit(S, w0, L, true, fl_del, sp.lim_del, tmed, vlo, vhi, at_start, at_end, S.FD, counters);
        ld = L;
      }
      if (sum >= S.tdup[L]) {
        if (lu == L - 1) { const uint32_t v = S.FU[grown]; wu = v > wu ? v : wu; } else wu = kUnmarked;
        if (wu > (uint32_t)L) wu = lane_hit(S, w0, L, false, ce_dup, sp.lim_dup, tmed, vlo, vhi, at_start, at_end, S.FU, counters);
        lu = L;
      }
    }
  }
  __syncthreads();
  // ---- merge the tile's marks into HBM (halo bins are shared with the neighbouring tiles) ----
  for (int e = vlo + threadIdx.x; e < vhi; e += kThreads) {
    const uint32_t d = S.FD[e], u = S.FU[e];
    if (d != kUnmarked) atomicMin(&first_del[lo + e], d);
    if (u != kUnmarked) atomicMin(&first_dup[lo + e], u);
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
  hipLaunchKernelGGL(k_minmax_f32, dim3(grid_for(nb, kThreads * 16)), dim3(kThreads), 0, stream, x, mask, nb, use_abs, center, mm);
}
void launch_hist_f32(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, double ymin, uint32_t* hist,
                     uint32_t np, hipStream_t stream) {
  const int use_lds = np <= kLdsBins;
  const size_t lds = use_lds ? (size_t)np * 4 : 0;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // few, long-lived workgroups: every one flushes np counters at the end
  int grid = grid_for(nb, kThreads * kHistRun * 8);
  if (grid > 128) grid = 128;
  hipLaunchKernelGGL(k_hist_f32, dim3(grid), dim3(kThreads), lds, stream, x, mask, nb, use_abs, center, ymin, hist, np, use_lds);
}
void launch_rsi_scan(const float* T, const int32_t* medint, const ScanParams& sp, const double* thr_del, const double* thr_dup,
                     uint32_t* first_del, uint32_t* first_dup, uint32_t* counters, hipStream_t stream) {
  const int halo = sp.Lmax / 2 + 1;
  const size_t count = (size_t)kScanTile + 2 * halo;
  const size_t lds = ((count + 1) + 2 * (size_t)(sp.Lmax + 1) + kThreads) * sizeof(double) + count * 4 * 4 + 2 * (count + 1) * 4;
  const int grid = (int)((sp.nb + kScanTile - 1) / kScanTile);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rsi_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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
