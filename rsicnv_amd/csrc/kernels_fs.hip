// kernels_fs.hip -- filterstatus' per-level sums on the device (rsi.cpp:948-1047, App. A Q13).
//
// The reference accumulates, for every status level, the transformed bin values of that level IN FLOAT, in index order:
// s <- fl(s + x).  The result depends on the order, so it cannot be a tree reduction; it can still be parallel, because
// within one binade of s (2^k <= s < 2^(k+1), unit u = 2^(k-23)) an addition is an INTEGER step: with s = S u and
// x = X u + r (0 <= r < u), fl(s + x) = (S + X + c) u where c = 0 / 1 for r below / above u/2 and, on a tie, whatever
// makes the result even.  So every element is a function S -> S + a[S & 1] with two small constants, such functions
// compose into functions of the same form, and the composition over a chunk of bins is an ordered reduction.  The binade
// the sum is in when it reaches a chunk is not known in advance, but it is known to within one: the float sum of at
// most 2^22 non-negative terms stays within 15 % of their exact sum, so two candidate binades per chunk cover it.  A
// last, sequential pass over the CHUNKS applies each chunk's function in one step when the sum stays inside the
// candidate binade (values are >= 0: if it is inside at the end it never left), and walks the chunk's bins one by one
// when it does not -- the ~25 binade crossings of a chromosome, and whatever precedes the first unmarked bin.
//
// The unmarked level (almost every bin) goes that way.  The marked levels hold few bins each: the marked bins are
// compacted in order, and one thread per level walks the compact list.
#include "kernels.h"
#include "device_util.h"

namespace rsik {

namespace {

constexpr int kThreads = 256;
constexpr int kPerThread = 8;
constexpr int kChunk = kThreads * kPerThread;   // bins per chunk

struct StepFn { long long a0, a1; };            // S -> S + (S even ? a0 : a1)
// An increment of 2^24 units or more takes the sum out of its binade whatever it is exactly: increments saturate at 2^40, so
// that no chain of compositions can overflow (the parity of a saturated increment is never used).
constexpr long long kStepSat = 1ll << 40;
__device__ inline StepFn fs_compose(StepFn f, StepFn g) {   // first f, then g
  StepFn h;
  h.a0 = f.a0 + ((f.a0 & 1) ? g.a1 : g.a0);                 // S even: S + f.a0 has the parity of f.a0
  h.a1 = f.a1 + (((1 + f.a1) & 1) ? g.a1 : g.a0);           // S odd
  h.a0 = h.a0 < kStepSat ? h.a0 : kStepSat;
  h.a1 = h.a1 < kStepSat ? h.a1 : kStepSat;
  return h;
}
// Ordered (non-commutative) inclusive composition over the workgroup's 256 functions: thread t gets f_0 ; f_1 ; ... ; f_t.
// Inside a wave the six steps are DPP moves (row shifts, then lane 15 / 31 into the rows above: the lanes a step does not reach
// receive {0, 0}, the identity), the four wave totals meet in LDS: ONE barrier where the LDS ladder took eight.  The caller
// keeps a barrier between the reads of s_wt here and the next call.
template <int CTRL, int ROWS, bool BOUND>
__device__ inline StepFn dpp_move_fn(StepFn f) { return StepFn{dpp_move_i64<CTRL, ROWS, BOUND>(f.a0), dpp_move_i64<CTRL, ROWS, BOUND>(f.a1)}; }
__device__ inline StepFn block_incl_scan_fn(StepFn x, StepFn* s_wt /* [kThreads / 64] */) {
  x = fs_compose(dpp_move_fn<0x111, 0xF, true>(x), x);
  x = fs_compose(dpp_move_fn<0x112, 0xF, true>(x), x);
  x = fs_compose(dpp_move_fn<0x114, 0xF, true>(x), x);
  x = fs_compose(dpp_move_fn<0x118, 0xF, true>(x), x);
  x = fs_compose(dpp_move_fn<0x142, 0xA, false>(x), x);
  x = fs_compose(dpp_move_fn<0x143, 0xC, false>(x), x);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) s_wt[wave] = x;
  __syncthreads();
  StepFn pre = {0, 0};
  for (int w = 0; w < wave; ++w) pre = fs_compose(pre, s_wt[w]);
  return fs_compose(pre, x);
}
// the step of adding x (a non-negative finite float) while the sum is in the binade with unit 2^ue
__device__ inline StepFn fs_step(float x, int ue) {
  // x = m 2^e exactly (m a 24-bit integer, or 0): in units of 2^ue it is m 2^(e-ue), integer part X, remainder r
  const uint32_t bits = __float_as_uint(x);
  const int ex = (int)((bits >> 23) & 0xff);
  unsigned long long m = ex ? ((bits & 0x7fffffu) | 0x800000u) : (bits & 0x7fffffu);
  const int e = (ex ? ex : 1) - 150;                        // x = m * 2^e
  const int sh = e - ue;
  StepFn f;
  if (m == 0) { f.a0 = f.a1 = 0; return f; }
  if (sh >= 0) { const long long X = (long long)(m << (sh > 16 ? 16 : sh)); f.a0 = f.a1 = X; return f; }   // a multiple of the unit (2^24 units and more: out of the binade anyway)
  const int d = -sh;                                        // X = m >> d, r = low d bits
  if (d > 25) { f.a0 = f.a1 = 0; return f; }                // less than a quarter of the unit: rounds away (m < 2^24)
  const long long X = (long long)(m >> d);
  const unsigned long long r = m & ((1ull << d) - 1), half = 1ull << (d - 1);
  if (r > half) { f.a0 = f.a1 = X + 1; }
  else if (r < half) { f.a0 = f.a1 = X; }
  else { f.a0 = X + (X & 1); f.a1 = X + ((X + 1) & 1); }    // tie: to even
  return f;
}

// Exclusive prefixes over the chunks' sums and marked counts, by one workgroup (the launch's last one to finish: the values were
// written by the others with write-through stores and are read past this CU's cache).
__device__ inline void fs_chunk_scan_block(double* __restrict__ csum, int32_t* __restrict__ cmark, int nchunks, int32_t* __restrict__ total_marked) {
  __shared__ double s_sd[kThreads / 64];
  __shared__ int s_sm[kThreads / 64];
  double cd = 0.0;
  int cm = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t0 = 0; t0 < nchunks; t0 += kThreads) {
    const int t = t0 + (int)threadIdx.x;
    const double v = t < nchunks ? __longlong_as_double((long long)ld_cg(reinterpret_cast<const unsigned long long*>(csum) + t)) : 0.0;
    const int m = t < nchunks ? ld_cg(cmark + t) : 0;
    double id = v; int im = m;   // inclusive wave scans (the double sums are bounds for the candidate binades, not results)
    for (int d = 1; d < 64; d <<= 1) { const double ud = __shfl_up(id, d); const int um = __shfl_up(im, d); if (lane >= d) { id += ud; im += um; } }
    __syncthreads();   // the previous batch's totals have been read
    if (lane == 63) { s_sd[wave] = id; s_sm[wave] = im; }
    __syncthreads();
    double pd = 0.0, td = 0.0; int pm = 0, tm = 0;
    for (int w = 0; w < kThreads / 64; ++w) { if (w < wave) { pd += s_sd[w]; pm += s_sm[w]; } td += s_sd[w]; tm += s_sm[w]; }
    if (t < nchunks) {   // written through, as the values they replace were (one dirty copy of a line per launch)
      st_cg(reinterpret_cast<unsigned long long*>(csum) + t, (unsigned long long)__double_as_longlong(cd + pd + id - v));
      st_cg(reinterpret_cast<unsigned int*>(cmark) + t, (unsigned int)(cm + pm + im - m));
    }
    cd += td; cm += tm;
  }
  if (threadIdx.x == 0) st_cg(reinterpret_cast<unsigned int*>(total_marked), (unsigned int)cm);
}

// Pass 1: per chunk, the exact-enough double sum and the count of the unmarked bins, the count of the marked ones; the last
// workgroup to finish turns them into exclusive prefixes (a launch of its own until round 5).
__global__ __launch_bounds__(kThreads) void k_fs_chunk_sums(const float* __restrict__ T, const int32_t* __restrict__ status, int64_t nb,
                                                            double* __restrict__ csum, int32_t* __restrict__ cmark,
                                                            int32_t* __restrict__ total /* [0]: marked bins; [1]: values the integer-step form cannot take */,
                                                            unsigned int* __restrict__ counter) {
  __shared__ double s_d[kThreads / 64];
  __shared__ int s_m[kThreads / 64];
  const int64_t i0 = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kPerThread;
  double acc = 0.0;
  int marked = 0;
  bool odd = false;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t i = i0 + k;
    if (i < nb) {
      const float x = T[i];
      if (!(x >= 0.0f && x <= 3.0e38f)) odd = true;   // negative, NaN or infinite (marked or not: the per-level walks add them too)
      if (status[i] == 0) acc += (double)x; else ++marked;
    }
  }
  if (odd) atomicOr(reinterpret_cast<unsigned int*>(&total[1]), 1u);
  for (int d = 32; d >= 1; d >>= 1) { acc += __shfl_xor(acc, d); marked += __shfl_xor(marked, d); }
  if ((threadIdx.x & 63) == 0) { s_d[threadIdx.x >> 6] = acc; s_m[threadIdx.x >> 6] = marked; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; ++w) { acc += s_d[w]; marked += s_m[w]; }
    st_cg(reinterpret_cast<unsigned long long*>(csum) + blockIdx.x, (unsigned long long)__double_as_longlong(acc));
    st_cg(reinterpret_cast<unsigned int*>(cmark) + blockIdx.x, (unsigned int)marked);
  }
  if (!last_block_done(counter)) return;
  fs_chunk_scan_block(csum, cmark, (int)gridDim.x, total);
}

// Pass 3: per chunk, the step functions of its unmarked bins for the two candidate binades of the sum that reaches it, and
// its marked bins appended, in order, to the compact list.
struct ChunkFn { int ue0; StepFn f0, f1; };   // unit exponents ue0 and ue0 + 1
__global__ __launch_bounds__(kThreads) void k_fs_chunk_fns(const float* __restrict__ T, const int32_t* __restrict__ status, int64_t nb,
                                                           const double* __restrict__ cpre, const int32_t* __restrict__ mpre,
                                                           ChunkFn* __restrict__ fns, int32_t* __restrict__ clist_s,
                                                           float* __restrict__ clist_t, int32_t clist_cap, int Lmax,
                                                           unsigned int* __restrict__ level_count) {
  __shared__ StepFn s_f[2][kThreads];
  __shared__ int s_cnt[kThreads];
  extern __shared__ unsigned int s_lev[];   // [2 Lmax + 1]: this chunk's marked bins per level
  for (int e = threadIdx.x; e < 2 * Lmax + 1; e += kThreads) s_lev[e] = 0;
  __syncthreads();
  // candidate binades: the float sum that reaches the chunk lies within 15 % of the exact prefix P (at most 2^22 terms, each
  // addition off by at most 2^-24 of the sum); the binade of 0.85 P and the one above it cover [0.85 P, 1.15 P]
  const double P = cpre[blockIdx.x];
  int ue0 = -200;
  if (P > 0.0) { int ex; (void)frexp(P * 0.85, &ex); ue0 = (ex - 1) - 23; }   // 0.85 P in [2^(ex-1), 2^ex)
  const int64_t i0 = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * kPerThread;
  StepFn f0 = {0, 0}, f1 = {0, 0};
  int nm = 0;
  int ms[kPerThread]; float mt[kPerThread];
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) {
    const int64_t i = i0 + k;
    ms[k] = 0; mt[k] = 0.0f;
    if (i < nb) {
      const int st = status[i];
      const float x = T[i];
      if (st == 0) { f0 = fs_compose(f0, fs_step(x, ue0)); f1 = fs_compose(f1, fs_step(x, ue0 + 1)); }
      else {
        ms[nm] = st; mt[nm] = x; ++nm;
        if (st >= -Lmax && st <= Lmax) atomicAdd(&s_lev[st + Lmax], 1u);   // which levels exist at all
      }
    }
  }
  {   // ordered composition over the 256 threads' functions, one scan per candidate binade: the last thread holds the chunk's function
    const StepFn c0 = block_incl_scan_fn(f0, s_f[0]);
    const StepFn c1 = block_incl_scan_fn(f1, s_f[1]);
    if (threadIdx.x == kThreads - 1) { fns[blockIdx.x].ue0 = ue0; fns[blockIdx.x].f0 = c0; fns[blockIdx.x].f1 = c1; }
  }
  s_cnt[threadIdx.x] = nm;
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * Lmax + 1; e += kThreads) { const unsigned int c = s_lev[e]; if (c) atomicAdd(&level_count[e], c); }
  int before = 0;
  for (int t = 0; t < (int)threadIdx.x; ++t) before += s_cnt[t];
  const int base = mpre[blockIdx.x] + before;
#pragma unroll
  for (int k = 0; k < kPerThread; ++k) if (k < nm && base + k < clist_cap) { clist_s[base + k] = ms[k]; clist_t[base + k] = mt[k]; }
}

// Pass 4, one launch.
//  * Workgroups 0 .. ceil((2 Lmax + 1) / 4) - 1: one WAVE per marked level.  The wave streams the compact list, 64 entries a
//    step, and keeps the values of ITS level in order without a barrier: ballot of the matches, a lane's slot is the number
//    of matching lanes below it, values into the wave's LDS queue; lane 0 adds the queue to the level's float sum whenever
//    it fills up.  A level no bin carries (most of them) returns at once.
//  * The last workgroup carries the unmarked level's float sum over the chunks, 256 chunk records at a time: one record per
//    thread, an ordered scan of the functions that fit the sum's binade, and the first chunk that cannot be taken in one step (a
//    binade crossing, or the start of the sum) is a count (a serial walk over the records by one lane took 0.2 us per chunk:
//    240 of the kernel's 340 us on a 250 Mb chromosome).  The workgroup then stages that chunk's bins in LDS, every thread
//    composes the step functions of its eight bins for the sum's current binade and the next, and the same scan finds the eight
//    bins where the sum actually crosses; lane 0 adds those one by one.
// out: [2 Lmax + 1] float sums then [2 Lmax + 1] int counts (index = level + Lmax); the workgroup that finishes last copies
// them to mapped host memory.  A count of -1 at the unmarked level tells the host to do the sums itself (more marked bins
// than the compact list holds, or a negative / non-finite value: the integer-step argument needs x >= 0).
__device__ inline bool fs_apply(float& s, int ue, StepFn f) {   // one step of a function valid for unit exponent ue; false: not applicable
  const uint32_t bits = __float_as_uint(s);
  const int ex = (int)((bits >> 23) & 0xff);
  if (ex == 0 || ex == 255 || ex - 150 != ue) return false;
  const long long S = (long long)((bits & 0x7fffffu) | 0x800000u);
  const long long S2 = S + ((S & 1) ? f.a1 : f.a0);
  if (S2 >= (1ll << 24)) return false;                          // the sum would leave the binade
  s = __uint_as_float(((uint32_t)ex << 23) | ((uint32_t)S2 & 0x7fffffu));
  return true;
}
__global__ __launch_bounds__(kThreads) void k_fs_level_sums(const float* __restrict__ T, const int32_t* __restrict__ status, int64_t nb,
                                                            int nchunks, const ChunkFn* __restrict__ fns, const int32_t* __restrict__ clist_s,
                                                            const float* __restrict__ clist_t, const int32_t* __restrict__ total_marked,
                                                            const unsigned int* __restrict__ level_count, int32_t clist_cap, int Lmax,
                                                            float* __restrict__ out, unsigned int* __restrict__ counter, void* host_copy) {
  __shared__ __align__(16) float s_x[kChunk];
  __shared__ StepFn s_ta[kThreads], s_tb[kThreads];
  __shared__ int s_t0, s_wcnt[kThreads / 64];
  __shared__ unsigned int s_sbits, s_cur;
  __shared__ StepFn s_wt[kThreads / 64];
  const int nlev = 2 * Lmax + 1;
  unsigned int* out_bits = reinterpret_cast<unsigned int*>(out);
  unsigned int* out_cnt = out_bits + nlev;
  const int M = total_marked[0];
  const bool bad = total_marked[1] != 0 || M > clist_cap;
  if (blockIdx.x == gridDim.x - 1) {
    // The sum walks the chunk records kThreads at a time: every thread holds one record, the ordered composition (a scan) of the
    // functions that fit the sum's binade gives every thread the sum AFTER its chunk; the results are monotone, so the first
    // chunk that takes the sum out of its binade (or whose candidate binades do not fit: a saturated increment stands in for it)
    // is a count.  That chunk goes through the bin-by-bin machinery below, and the walk continues behind it.
    if (threadIdx.x == 0) s_cur = 0u;   // 0.0f
    for (int b0 = 0; b0 < nchunks; b0 += kThreads) {
      const int bn = nchunks - b0 < kThreads ? nchunks - b0 : kThreads;
      ChunkFn R;
      R.ue0 = 0; R.f0 = StepFn{0, 0}; R.f1 = StepFn{0, 0};
      if ((int)threadIdx.x < bn) R = fns[b0 + threadIdx.x];
      __syncthreads();   // s_cur
      int c = 0;
      while (c < bn) {
        {
          const uint32_t bits = s_cur;
          const int ex = (int)((bits >> 23) & 0xff);
          if (ex != 0 && ex != 255) {
            const int d = (ex - 150) - R.ue0;
            StepFn F = {0, 0};
            if ((int)threadIdx.x >= c && (int)threadIdx.x < bn) F = d == 0 ? R.f0 : (d == 1 ? R.f1 : StepFn{kStepSat, kStepSat});
            F = block_incl_scan_fn(F, s_wt);
            const long long S = (long long)((bits & 0x7fffffu) | 0x800000u);
            const long long St = S + ((S & 1) ? F.a1 : F.a0);
            const unsigned long long okm = __ballot(St < (1ll << 24));
            if ((threadIdx.x & 63) == 0) s_wcnt[threadIdx.x >> 6] = __popcll(okm);
            __syncthreads();
            int tfail = 0;   // chunks taken in one step each (threads below c and from bn on carry the identity)
            for (int w = 0; w < kThreads / 64; ++w) tfail += s_wcnt[w];
            if (tfail > 0 && (int)threadIdx.x == tfail - 1) s_cur = ((uint32_t)ex << 23) | ((uint32_t)St & 0x7fffffu);
            __syncthreads();
            c = tfail < bn ? tfail : bn;
          }
        }
        if (c >= bn) break;
        float s = __uint_as_float(s_cur);
        // chunk b0 + c crosses a binade (or starts the sum): its unmarked values into LDS (marked ones as -1), per-thread
        // step functions for the sum's binade and the one above.  Then, in rounds: an ordered parallel composition (scan) of
        // the 256 functions from the first thread not yet consumed gives every thread the sum AFTER its bins as an integer; the
        // results are monotone, so the first thread whose result leaves the binade is a count; lane 0 adds that thread's eight
        // bins one by one in float (the crossing itself), and the next round continues behind it in the new binade.
        int ue = 0;   // set when the functions are built
        const int64_t i0 = (int64_t)(b0 + c) * kChunk + (int64_t)threadIdx.x * kPerThread;
#pragma unroll
        for (int k = 0; k < kPerThread; ++k) {
          const int64_t i = i0 + k;
          const bool use = i < nb && status[i] == 0;
          s_x[threadIdx.x * kPerThread + k] = use ? T[i] : -1.0f;
        }
        bool staged = false;
        if (threadIdx.x == 0) { s_sbits = __float_as_uint(s); s_t0 = 0; }
        __syncthreads();
        // The chromosome's first chunk takes the sum from zero through a dozen binades (one round of the machinery below for
        // each): lane 0 adds its bins one after the other instead -- the loop itself, 2048 dependent additions, about one round's time.
        if (b0 + c == 0) {
          if (threadIdx.x == 0) {
            float sv = __uint_as_float(s_sbits);
            for (int j = 0; j < kChunk; j += 4) {
              const float4 x = *reinterpret_cast<const float4*>(&s_x[j]);
              sv = x.x >= 0.0f ? sv + x.x : sv; sv = x.y >= 0.0f ? sv + x.y : sv;
              sv = x.z >= 0.0f ? sv + x.z : sv; sv = x.w >= 0.0f ? sv + x.w : sv;
            }
            s_sbits = __float_as_uint(sv); s_t0 = kThreads;
          }
          __syncthreads();
        }
        while (true) {
          if (threadIdx.x == 0) {   // a sum that is not a normal number yet (the start: zeros) takes bins one by one
            float sv = __uint_as_float(s_sbits);
            int t0 = s_t0;
            while (t0 < kThreads) {
              const int ex = (int)((__float_as_uint(sv) >> 23) & 0xff);
              if (ex != 0 && ex != 255) break;
#pragma unroll
              for (int k = 0; k < kPerThread; ++k) { const float x = s_x[t0 * kPerThread + k]; sv = x >= 0.0f ? sv + x : sv; }
              ++t0;
            }
            s_sbits = __float_as_uint(sv); s_t0 = t0;
          }
          __syncthreads();
          const int t0 = s_t0;
          if (t0 >= kThreads) break;
          const uint32_t bits = s_sbits;
          const int ex = (int)((bits >> 23) & 0xff);
          if (!staged || (ex - 150 != ue && ex - 150 != ue + 1)) {   // (re)build the functions for the binade the sum is in now
            ue = ex - 150;
            StepFn fa = {0, 0}, fb = {0, 0};
#pragma unroll
            for (int k = 0; k < kPerThread; ++k) {
              const float x = s_x[threadIdx.x * kPerThread + k];
              if (x >= 0.0f) { fa = fs_compose(fa, fs_step(x, ue)); fb = fs_compose(fb, fs_step(x, ue + 1)); }
            }
            s_ta[threadIdx.x] = fa; s_tb[threadIdx.x] = fb;
            staged = true;
          }
          const StepFn* fn = ex - 150 == ue ? s_ta : s_tb;
          StepFn F = {0, 0};
          if ((int)threadIdx.x >= t0) F = fn[threadIdx.x];   // own entry: no barrier needed before reading it
          F = block_incl_scan_fn(F, s_wt);
          const long long S = (long long)((bits & 0x7fffffu) | 0x800000u);
          const long long St = S + ((S & 1) ? F.a1 : F.a0);
          const bool ok = St < (1ll << 24);
          const unsigned long long okm = __ballot(ok);
          if ((threadIdx.x & 63) == 0) s_wcnt[threadIdx.x >> 6] = __popcll(okm);
          __syncthreads();
          int tfail = 0;
          for (int w = 0; w < kThreads / 64; ++w) tfail += s_wcnt[w];
          if ((int)threadIdx.x == tfail - 1) s_sbits = ((uint32_t)ex << 23) | ((uint32_t)St & 0x7fffffu);   // threads below t0 carry S itself
          __syncthreads();
          if (tfail >= kThreads) break;
          if (threadIdx.x == 0) {
            float sv = __uint_as_float(s_sbits);
#pragma unroll
            for (int k = 0; k < kPerThread; ++k) { const float x = s_x[tfail * kPerThread + k]; sv = x >= 0.0f ? sv + x : sv; }
            s_sbits = __float_as_uint(sv); s_t0 = tfail + 1;
          }
          __syncthreads();
        }
        if (threadIdx.x == 0) s_cur = s_sbits;
        ++c;
        __syncthreads();
      }
    }
    if (threadIdx.x == 0) { st_cg(&out_bits[Lmax], s_cur); st_cg(&out_cnt[Lmax], bad ? 0xffffffffu : (unsigned int)(nb - M)); }
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = (int)blockIdx.x * (kThreads / 64) + wave, l = li - Lmax;
    if (li < nlev && l != 0) {
      float* q = s_x + wave * (kChunk / (kThreads / 64));   // the wave's queue: 512 floats
      constexpr int kQ = kChunk / (kThreads / 64);
      float s = 0.0f;
      int cnt = 0, qn = 0;
      if (!bad && level_count[li] != 0) {
        int nxt[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int e = 64 * j + lane; nxt[j] = e < M ? clist_s[e] : 0; }   // 0 is never a marked level
        for (int k0 = 0; k0 < M; k0 += 64 * 8) {   // eight loads in flight per lane, one batch ahead of the eight 64-entry steps
          int key[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) key[j] = nxt[j];
#pragma unroll
          for (int j = 0; j < 8; ++j) { const int e = k0 + 64 * 8 + 64 * j + lane; nxt[j] = e < M ? clist_s[e] : 0; }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool hit = key[j] == l;
            const unsigned long long mask = __ballot(hit);
            if (mask) {
              if (qn + 64 > kQ) {   // drain: lane 0 adds the queued values in order
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) for (int jj = 0; jj < qn; ++jj) s += q[jj];
                qn = 0;
                __builtin_amdgcn_wave_barrier();
              }
              if (hit) q[qn + __popcll(mask & ((1ull << lane) - 1))] = clist_t[k0 + 64 * j + lane];
              const int m = __popcll(mask);
              qn += m; cnt += m;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) for (int j = 0; j < qn; ++j) s += q[j];
      }
      if (lane == 0) { st_cg(&out_bits[li], __float_as_uint(s)); st_cg(&out_cnt[li], (unsigned int)cnt); }
    }
  }
  if (!last_block_done(counter)) return;
  export_words(host_copy, out, (size_t)nlev * 8);
}

}  // namespace

size_t level_sums_head_bytes(int Lmax) { return (16 + (size_t)(2 * Lmax + 1) * 4 + 15) & ~size_t(15); }   // what must be zero before a launch
size_t level_sums_workspace_bytes(int64_t nb, int32_t clist_cap, int Lmax_cap) {
  const size_t nchunks = (size_t)((nb + kChunk - 1) / kChunk);
  return level_sums_head_bytes(Lmax_cap) + nchunks * (8 + sizeof(ChunkFn)) + ((nchunks * 4 + 15) & ~size_t(15)) + (size_t)clist_cap * 8 + 256;
}
void launch_level_sums(const float* T, const int32_t* status, int64_t nb, int Lmax, void* ws, int Lmax_cap, int32_t clist_cap, float* out,
                       unsigned int* counter, void* host_copy, hipStream_t stream) {
  const int nchunks = (int)((nb + kChunk - 1) / kChunk);
  unsigned char* p = static_cast<unsigned char*>(ws);
  int32_t* total = reinterpret_cast<int32_t*>(p);            // [0] marked bins, [1] flag
  unsigned int* level_count = reinterpret_cast<unsigned int*>(p + 16);
  p += level_sums_head_bytes(Lmax_cap);
  double* csum = reinterpret_cast<double*>(p); p += (size_t)nchunks * 8;
  ChunkFn* fns = reinterpret_cast<ChunkFn*>(p); p += (size_t)nchunks * sizeof(ChunkFn);
  int32_t* cmark = reinterpret_cast<int32_t*>(p); p += ((size_t)nchunks * 4 + 15) & ~size_t(15);
  int32_t* clist_s = reinterpret_cast<int32_t*>(p); p += (size_t)clist_cap * 4;
  float* clist_t = reinterpret_cast<float*>(p);
  RSI_LAUNCH(k_fs_chunk_sums, dim3(nchunks), dim3(kThreads), 0, stream, T, status, nb, csum, cmark, total, counter);   // (the counter is back at zero when the launch ends)
  RSI_ALLOW_FULL_LDS(k_fs_chunk_fns);   // (2 Lmax + 1) * 4 bytes: 80 KB at -m 1
  RSI_LAUNCH(k_fs_chunk_fns, dim3(nchunks), dim3(kThreads), (size_t)(2 * Lmax + 1) * 4, stream, T, status, nb, csum, cmark, fns, clist_s, clist_t, clist_cap, Lmax, level_count);
  RSI_LAUNCH(k_fs_level_sums, dim3((2 * Lmax + 1 + kThreads / 64 - 1) / (kThreads / 64) + 1), dim3(kThreads), 0, stream, T, status, nb, nchunks, fns, clist_s, clist_t, total, level_count,
                     clist_cap, Lmax, out, counter, host_copy);
}

}  // namespace rsik
