// kernels_base.hip -- per-base streaming kernels (gfx950): FASTA classification, GC table,
// GC rescale, cap + N-compaction + bin reduction.  All are HBM-bound integer/byte work
// (DESIGN.md section 4): coalesced 16-byte loads, bit-packed GC masks staged through LDS,
// LDS-privatised histograms with bank-spreading replicas, persistent grids sized to the chip.
//
// Built with -ffp-contract=off: the only floating-point here is the GC rescale, which must round
// exactly like the reference's x86-64 SSE2 build.
#include "kernels.h"

namespace rsik {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxGrid = 256 * 8;          // 256 CUs x 8 resident workgroups
constexpr int kGcWords = kTileBases / 64;  // 64 words per tile
constexpr int kGcLeft = 4;                 // margin words left of the tile (256 bits >= 201)
constexpr int kGcRight = 2;                // margin words right of the tile (128 bits >= 101)
constexpr int kGcLds = kGcLeft + kGcWords + kGcRight;   // 70

__device__ inline int lane_id() { return threadIdx.x & 63; }

// ------------------------------------------------------------------------------------------
// K1  fasta_classify: one thread per 16 bytes; 4 neighbouring lanes assemble one 64-bit word.
__global__ __launch_bounds__(kThreads) void k_fasta_classify(const uint8_t* __restrict__ fasta, int64_t n,
                                                             uint64_t* __restrict__ gcbits,
                                                             uint64_t* __restrict__ nbits, int64_t nwords) {
  const int64_t nthreads16 = nwords * 4;   // 16-byte groups to cover all words
  for (int64_t g = (int64_t)blockIdx.x * kThreads + threadIdx.x; g < nthreads16;
       g += (int64_t)gridDim.x * kThreads) {
    const int64_t base = g * 16;
    uint32_t w[4] = {0, 0, 0, 0};
    if (base + 16 <= n) {
      const uint4 v = *reinterpret_cast<const uint4*>(fasta + base);
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else if (base < n) {
      for (int j = 0; j < 16 && base + j < n; ++j) w[j >> 2] |= (uint32_t)fasta[base + j] << (8 * (j & 3));
    }
    uint32_t mg = 0, mn = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t c = (w[j >> 2] >> (8 * (j & 3))) & 0xffu;
      mg |= (uint32_t)(c == 'G' || c == 'C') << j;
      mn |= (uint32_t)(c == 'N') << j;
    }
    const int sub = (int)(g & 3);
    uint64_t vg = (uint64_t)mg << (16 * sub), vn = (uint64_t)mn << (16 * sub);
    // the 4 lanes of a word are adjacent lanes of one wave (g is contiguous in threadIdx.x)
    vg |= __shfl_xor(vg, 1); vg |= __shfl_xor(vg, 2);
    vn |= __shfl_xor(vn, 1); vn |= __shfl_xor(vn, 2);
    if (sub == 0) { gcbits[g >> 2] = vg; nbits[g >> 2] = vn; }
  }
}

// K1b n_transitions: run starts and (exclusive) ends of the N mask.
__global__ __launch_bounds__(kThreads) void k_n_transitions(const uint64_t* __restrict__ nbits, int64_t nwords,
                                                            uint64_t* __restrict__ list, uint32_t* __restrict__ count,
                                                            uint32_t cap) {
  for (int64_t w = (int64_t)blockIdx.x * kThreads + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * kThreads) {
    const uint64_t cur = nbits[w];
    const uint64_t prev_top = w > 0 ? (nbits[w - 1] >> 63) : 0;
    const uint64_t shifted = (cur << 1) | prev_top;
    uint64_t starts = cur & ~shifted, ends = ~cur & shifted;
    while (starts) {
      const int b = __ffsll((long long)starts) - 1;
      starts &= starts - 1;
      const uint32_t k = atomicAdd(count, 1u);
      if (k < cap) list[k] = ((uint64_t)(w * 64 + b) << 1);
    }
    while (ends) {
      const int b = __ffsll((long long)ends) - 1;
      ends &= ends - 1;
      const uint32_t k = atomicAdd(count, 1u);
      if (k < cap) list[k] = ((uint64_t)(w * 64 + b) << 1) | 1u;
    }
  }
}

// ------------------------------------------------------------------------------------------
// GC window counts.  The tile's GC words (with margins) are staged in LDS together with an
// exclusive prefix of their popcounts; rank(x) = #GC in [first staged bit, x).
struct GcTile {
  uint64_t word[kGcLds];
  uint32_t pre[kGcLds + 1];
};

__device__ inline void gc_tile_load(GcTile& t, const uint64_t* __restrict__ gcbits, int64_t nwords, int64_t tile_word0) {
  // wave 0 loads and scans; callers __syncthreads() afterwards
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    const int64_t gw = tile_word0 - kGcLeft + l;
    const uint64_t w = (gw >= 0 && gw < nwords) ? gcbits[gw] : 0;
    t.word[l] = w;
    uint32_t c = __popcll(w), incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d);
      if (l >= d) incl += up;
    }
    t.pre[l] = incl - c;
    const uint32_t total = __shfl(incl, 63);
    // remaining words 64..69 by lanes 0..5
    uint64_t w2 = 0;
    if (l < kGcLds - 64) {
      const int64_t gw2 = tile_word0 - kGcLeft + 64 + l;
      w2 = (gw2 >= 0 && gw2 < nwords) ? gcbits[gw2] : 0;
      t.word[64 + l] = w2;
    }
    uint32_t c2 = __popcll(w2), incl2 = c2;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      const uint32_t up = __shfl_up(incl2, d);
      if (l >= d) incl2 += up;
    }
    if (l < kGcLds - 64) t.pre[64 + l] = total + incl2 - c2;
    if (l == kGcLds - 64 - 1) t.pre[kGcLds] = total + incl2;
  }
}

// #GC in [first staged bit, first staged bit + rel)
__device__ inline uint32_t gc_rank(const GcTile& t, uint32_t rel) {
  const uint32_t k = rel >> 6, b = rel & 63;
  const uint64_t m = b ? (t.word[k] & ((1ull << b) - 1)) : 0;
  return t.pre[k] + (uint32_t)__popcll(m);
}

// Window GC count of base i with the reference's edge rules (gccontent.cpp:124-133, App. A Q1):
// lo = clamp(i-100, 0, n-202), window [lo, lo+200].
__device__ inline int gc_window(const GcTile& t, int64_t i, int64_t n, int64_t first_bit) {
  int64_t lo = i - 100;
  if (lo < 0) lo = 0;
  if (lo > n - 202) lo = n - 202;
  const uint32_t rel = (uint32_t)(lo - first_bit);
  return (int)(gc_rank(t, rel + 201) - gc_rank(t, rel));
}

// ------------------------------------------------------------------------------------------
// K2  gc_hist
constexpr int kGcRep = 8;   // LDS replicas per GC level, selected by lane & 7 (spreads hot levels over banks)

__global__ __launch_bounds__(kThreads) void k_gc_hist(const int32_t* __restrict__ depth,
                                                      const uint64_t* __restrict__ gcbits, int64_t n, int64_t nwords,
                                                      GcAccum* __restrict__ acc) {
  __shared__ GcTile gt;
  __shared__ unsigned long long s_sum[kGcLevels * kGcRep];
  __shared__ unsigned int s_cnt[kGcLevels * kGcRep];
  for (int e = threadIdx.x; e < kGcLevels * kGcRep; e += kThreads) { s_sum[e] = 0; s_cnt[e] = 0; }
  unsigned long long possum = 0, poscnt = 0;
  unsigned int neg = 0;
  const int rep = threadIdx.x & (kGcRep - 1);
  const int64_t ntiles = (n + kTileBases - 1) / kTileBases;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * kTileBases;
    __syncthreads();   // previous tile's readers are done with gt (and the zeroing above)
    gc_tile_load(gt, gcbits, nwords, base / 64);
    __syncthreads();
    const int64_t first_bit = base - kGcLeft * 64;
#pragma unroll
    for (int k = 0; k < kTileBases / (4 * kThreads); ++k) {
      const int64_t q = base + 4 * (int64_t)(k * kThreads + threadIdx.x);
      if (q >= n) continue;
      int v[4] = {0, 0, 0, 0};
      int cntv = 4;
      if (q + 4 <= n) {
        const int4 d = *reinterpret_cast<const int4*>(depth + q);
        v[0] = d.x; v[1] = d.y; v[2] = d.z; v[3] = d.w;
      } else {
        cntv = (int)(n - q);
        for (int j = 0; j < cntv; ++j) v[j] = depth[q + j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j >= cntv) break;
        const int g = gc_window(gt, q + j, n, first_bit);
        atomicAdd(&s_sum[g * kGcRep + rep], (unsigned long long)(long long)v[j]);
        atomicAdd(&s_cnt[g * kGcRep + rep], 1u);
        if (v[j] > 0) { possum += (unsigned long long)v[j]; poscnt += 1; }
        if (v[j] < 0) neg = 1;
      }
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < kGcLevels; g += kThreads) {
    unsigned long long s = 0, c = 0;
    for (int r = 0; r < kGcRep; ++r) { s += s_sum[g * kGcRep + r]; c += s_cnt[g * kGcRep + r]; }
    if (c) { atomicAdd(&acc->sum[g], s); atomicAdd(&acc->cnt[g], c); }
  }
  // wave reduction of the positive-depth sums, one atomic per wave
  for (int d = 32; d >= 1; d >>= 1) {
    possum += __shfl_xor(possum, d);
    poscnt += __shfl_xor(poscnt, d);
    neg |= __shfl_xor(neg, d);
  }
  if (lane_id() == 0) {
    if (poscnt) { atomicAdd(&acc->possum, possum); atomicAdd(&acc->poscnt, poscnt); }
    if (neg) atomicOr(&acc->negatives, 1u);
  }
}

// ------------------------------------------------------------------------------------------
// K3  gc_rescale (+ value histogram for the cap median)
constexpr int kValLds = 256;   // values below this are counted in LDS, [value][32 lane phases]

__device__ inline void value_hist_add(unsigned int* s_hist, uint32_t* __restrict__ ghist, ValueHistAux* aux, int v,
                                      int phase) {
  if (v >= 0 && v < kValLds) atomicAdd(&s_hist[v * 32 + phase], 1u);
  else if (v >= 0 && v < kHistValues) atomicAdd(&ghist[v], 1u);
  else if (v < 0) atomicOr(&aux->negatives, 1u);
  else { atomicAdd(&aux->big, 1ull); atomicMax(&aux->vmax, (unsigned int)v); }
}

template <bool ADJUST>
__global__ __launch_bounds__(kThreads) void k_gc_rescale(const int32_t* __restrict__ depth,
                                                         const uint64_t* __restrict__ gcbits, int64_t n,
                                                         int64_t nwords, const double* __restrict__ table,
                                                         double rdmean, int32_t* __restrict__ out,
                                                         uint32_t* __restrict__ ghist, ValueHistAux* __restrict__ aux) {
  __shared__ GcTile gt;
  __shared__ double s_table[kGcLevels];
  __shared__ unsigned int s_hist[kValLds * 32];
  for (int e = threadIdx.x; e < kValLds * 32; e += kThreads) s_hist[e] = 0;
  if (ADJUST) for (int e = threadIdx.x; e < kGcLevels; e += kThreads) s_table[e] = table[e];
  const int phase = threadIdx.x & 31;
  const int64_t ntiles = (n + kTileBases - 1) / kTileBases;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * kTileBases;
    const int64_t first_bit = base - kGcLeft * 64;
    if (ADJUST) {
      __syncthreads();
      gc_tile_load(gt, gcbits, nwords, base / 64);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTileBases / (4 * kThreads); ++k) {
      const int64_t q = base + 4 * (int64_t)(k * kThreads + threadIdx.x);
      if (q >= n) continue;
      int v[4] = {0, 0, 0, 0};
      int cntv = 4;
      if (q + 4 <= n) {
        const int4 d = *reinterpret_cast<const int4*>(depth + q);
        v[0] = d.x; v[1] = d.y; v[2] = d.z; v[3] = d.w;
      } else {
        cntv = (int)(n - q);
        for (int j = 0; j < cntv; ++j) v[j] = depth[q + j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j >= cntv) break;
        if (ADJUST) {
          const int g = gc_window(gt, q + j, n, first_bit);
          // RDA[k] = RD[i]*RDmean/GCRD[nGC] + 0.5, truncated to int (gccontent.cpp:89)
          v[j] = (int)((double)v[j] * rdmean / s_table[g] + 0.5);
        }
        value_hist_add(s_hist, ghist, aux, v[j], phase);
      }
      if (out) {
        if (cntv == 4) *reinterpret_cast<int4*>(out + q) = make_int4(v[0], v[1], v[2], v[3]);
        else for (int j = 0; j < cntv; ++j) out[q + j] = v[j];
      }
    }
  }
  __syncthreads();
  for (int v = threadIdx.x; v < kValLds; v += kThreads) {
    unsigned int c = 0;
    for (int p = 0; p < 32; ++p) c += s_hist[v * 32 + ((p + v) & 31)];
    if (c) atomicAdd(&ghist[v], c);
  }
}

// Tail of the 20-slice write-back (gccontent.cpp:156-175; App. A Q2/Q3).  One thread.
__global__ void k_gc_tail_fixup(const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits, int64_t n,
                                const double* __restrict__ table, double rdmean, int32_t* __restrict__ out,
                                uint32_t* __restrict__ ghist, ValueHistAux* __restrict__ aux) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int64_t S = n / 20, r = n - 20 * S;
  if (r == 0) return;
  auto hist_move = [&](int from, int to) {
    if (from == to) return;
    if (from >= 0 && from < kHistValues) atomicSub(&ghist[from], 1u);
    else if (from >= kHistValues) atomicAdd(&aux->big, (unsigned long long)-1ll);
    if (to >= 0 && to < kHistValues) atomicAdd(&ghist[to], 1u);
    else if (to >= kHistValues) { atomicAdd(&aux->big, 1ull); atomicMax(&aux->vmax, (unsigned int)to); }
    else atomicOr(&aux->negatives, 1u);
  };
  if (r >= 2) {
    int gtail = 0;   // fresh edge window [n-201, n-1]
    for (int64_t i = n - 201; i < n; ++i) gtail += (int)((gcbits[i >> 6] >> (i & 63)) & 1);
    for (int64_t k = 0; k < r; ++k) {
      const int nv = (int)((double)depth[20 * S + k] * rdmean / table[gtail] + 0.5);
      const int64_t idx = n - 201 + k;
      hist_move(out[idx], nv);
      out[idx] = nv;
    }
  }
  for (int64_t k = 0; k < r; ++k) {   // the last r bases keep their unadjusted depth
    const int64_t idx = 20 * S + k;
    hist_move(out[idx], depth[idx]);
    out[idx] = depth[idx];
  }
}

// ------------------------------------------------------------------------------------------
// K4  cap_compact_bin.  One workgroup per tile of TB bins (TB*m compacted bases) staged in LDS.
__device__ inline int upper_bound_i64(const int64_t* a, int n, int64_t key) {   // first index with a[idx] > key
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
  return lo;
}

__global__ __launch_bounds__(kThreads) void k_cap_compact_bin(
    const int32_t* __restrict__ src, int64_t n, const int64_t* __restrict__ cbreak, const int64_t* __restrict__ cum,
    int nreg, int64_t ncompact, int32_t capval, int m, int TB, int vr /* LDS histogram value range, power of two */,
    int32_t* __restrict__ rdc, int32_t* __restrict__ binmed, int64_t* __restrict__ binsum,
    uint32_t* __restrict__ res_hist, BinAccum* __restrict__ acc) {
  extern __shared__ __align__(16) unsigned char smem[];
  int32_t* s_val = reinterpret_cast<int32_t*>(smem);                       // TB*m values (padded to 4)
  const int tile_elems = TB * m;
  const int tile_pad = (tile_elems + 3) & ~3;
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + (size_t)tile_pad * 4);   // [vr][32]
  for (int e = threadIdx.x; e < vr * kResClasses; e += kThreads) s_hist[e] = 0;

  const int64_t lim31 = (ncompact / 31) * 31;
  const int64_t nb = ncompact / m;
  const int64_t ntiles = (ncompact + tile_elems - 1) / tile_elems;
  unsigned long long t_sum = 0, t_sqlo = 0, t_sqhi = 0;
  const int parts = kThreads / TB;          // threads cooperating on one bin
  const int kth = (m + 1) / 2;              // rank of the median, m odd (rsi.cpp:2061)

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t P0 = tile * tile_elems;
    const int64_t P1 = (P0 + tile_elems < ncompact) ? P0 + tile_elems : ncompact;
    __syncthreads();   // s_val free (and s_hist zeroed on the first trip)
    // ---- stage the tile: contiguous source segments between removed regions ----
    int k = upper_bound_i64(cbreak, nreg, P0);   // regions already cut out before P0
    int64_t seg = P0;
    while (seg < P1) {
      const int64_t nxt = (k < nreg && cbreak[k] < P1) ? cbreak[k] : P1;
      const int64_t len = nxt - seg;
      if (len > 0) {
        const int64_t soff = seg + cum[k];
        const int mis = (int)(soff & 3);
        const int64_t a = soff - mis;
        const int dst = (int)(seg - P0);
        for (int64_t i4 = threadIdx.x; i4 * 4 < len + mis; i4 += kThreads) {
          const int64_t s = a + 4 * i4;
          int v[4];
          if (s + 4 <= n) {
            const int4 d = *reinterpret_cast<const int4*>(src + s);
            v[0] = d.x; v[1] = d.y; v[2] = d.z; v[3] = d.w;
          } else {
            for (int j = 0; j < 4; ++j) v[j] = (s + j < n) ? src[s + j] : 0;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t e = 4 * i4 + j - mis;
            if (e < 0 || e >= len) continue;
            int x = v[j];
            if (x > capval) x = capval;
            s_val[dst + e] = x;
            const int64_t p = seg + e;                       // compacted index
            const int cls = p < lim31 ? (int)((uint32_t)p % 31u) : 31;   // n < 2^31 (checked by the caller)
            if (x >= 0 && x < vr) atomicAdd(&s_hist[x * kResClasses + cls], 1u);
            else if (x >= 0 && x < kHistValues) atomicAdd(&res_hist[(size_t)x * kResClasses + cls], 1u);
            else { atomicAdd(&acc->big, 1ull); atomicMax(&acc->vmax, (unsigned int)x); }
            const unsigned long long ux = (unsigned long long)(long long)x;
            const unsigned long long sq = ux * ux;
            t_sum += ux; t_sqlo += sq & 0xffffffffull; t_sqhi += sq >> 32;
          }
        }
      }
      seg = nxt;
      if (k < nreg && cbreak[k] == nxt) ++k;
    }
    __syncthreads();
    // ---- compacted, capped depth back to HBM (16-byte stores; P0 is a multiple of 4) ----
    const int cnt = (int)(P1 - P0);
    for (int i4 = threadIdx.x; i4 * 4 < cnt; i4 += kThreads) {
      if (i4 * 4 + 4 <= cnt) *reinterpret_cast<int4*>(rdc + P0 + 4 * i4) = *reinterpret_cast<const int4*>(s_val + 4 * i4);
      else for (int j = 4 * i4; j < cnt; ++j) rdc[P0 + j] = s_val[j];
    }
    // ---- per-bin exact median (order statistic kth) and sum: `parts` threads per bin ----
    const int b_local = threadIdx.x / parts, part = threadIdx.x % parts;
    const int64_t b = tile * TB + b_local;
    const bool active = b < nb && b_local < TB;
    const int32_t* x = s_val + b_local * m;
    int lo = 0x7fffffff, hi = (int)0x80000000;
    long long ssum = 0;
    if (active) for (int j = part; j < m; j += parts) { const int v = x[j]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; ssum += v; }
    for (int d = 1; d < parts; d <<= 1) {
      const int olo = __shfl_xor(lo, d), ohi = __shfl_xor(hi, d);
      const long long os = __shfl_xor(ssum, d);
      lo = olo < lo ? olo : lo; hi = ohi > hi ? ohi : hi; ssum += os;
    }
    // bisection on the value: smallest v with #{x <= v} >= kth
    while (__any(active && lo < hi)) {
      const int mid = (int)(((long long)lo + (long long)hi) >> 1);
      int c = 0;
      if (active && lo < hi) for (int j = part; j < m; j += parts) c += x[j] <= mid;
      for (int d = 1; d < parts; d <<= 1) c += __shfl_xor(c, d);
      if (active && lo < hi) { if (c >= kth) hi = mid; else lo = mid + 1; }
    }
    if (active && part == 0) { binmed[b] = lo; binsum[b] = ssum; }
  }
  __syncthreads();
  // ---- flush the LDS histogram ----
  for (int e = threadIdx.x; e < vr * kResClasses; e += kThreads) {
    const unsigned int c = s_hist[e];
    if (c) atomicAdd(&res_hist[e], c);   // same [value][class] layout as the global histogram
  }
  for (int d = 32; d >= 1; d >>= 1) {
    t_sum += __shfl_xor(t_sum, d); t_sqlo += __shfl_xor(t_sqlo, d); t_sqhi += __shfl_xor(t_sqhi, d);
  }
  if (lane_id() == 0 && (t_sum | t_sqlo | t_sqhi)) {
    atomicAdd(&acc->sum, t_sum); atomicAdd(&acc->sq_lo, t_sqlo); atomicAdd(&acc->sq_hi, t_sqhi);
  }
}

inline int grid_for(int64_t items, int per_block) {
  int64_t g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

}  // namespace

void launch_fasta_classify(const uint8_t* fasta, int64_t n, uint64_t* gcbits, uint64_t* nbits, int64_t nwords,
                           hipStream_t stream) {
  hipLaunchKernelGGL(k_fasta_classify, dim3(grid_for(nwords * 4, kThreads)), dim3(kThreads), 0, stream, fasta, n, gcbits,
                     nbits, nwords);
}
void launch_n_transitions(const uint64_t* nbits, int64_t nwords, uint64_t* list, uint32_t* count, uint32_t cap,
                          hipStream_t stream) {
  hipLaunchKernelGGL(k_n_transitions, dim3(grid_for(nwords, kThreads)), dim3(kThreads), 0, stream, nbits, nwords, list,
                     count, cap);
}
void launch_gc_hist(const int32_t* depth, const uint64_t* gcbits, int64_t n, GcAccum* acc, hipStream_t stream) {
  hipLaunchKernelGGL(k_gc_hist, dim3(grid_for(n, kTileBases)), dim3(kThreads), 0, stream, depth, gcbits, n, n / 64 + 1, acc);
}
void launch_gc_rescale(const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table, double rdmean,
                       int adjust, int32_t* out, uint32_t* hist, ValueHistAux* aux, hipStream_t stream) {
  const dim3 g(grid_for(n, kTileBases)), b(kThreads);
  if (adjust) hipLaunchKernelGGL(k_gc_rescale<true>, g, b, 0, stream, depth, gcbits, n, n / 64 + 1, table, rdmean, out, hist, aux);
  else hipLaunchKernelGGL(k_gc_rescale<false>, g, b, 0, stream, depth, gcbits, n, n / 64 + 1, table, rdmean, out, hist, aux);
}
void launch_gc_tail_fixup(const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table, double rdmean,
                          int32_t* out, uint32_t* hist, ValueHistAux* aux, hipStream_t stream) {
  hipLaunchKernelGGL(k_gc_tail_fixup, dim3(1), dim3(64), 0, stream, depth, gcbits, n, table, rdmean, out, hist, aux);
}

void launch_cap_compact_bin(const int32_t* src, int64_t n, const int64_t* cbreak, const int64_t* cum, int nreg,
                            int64_t ncompact, int32_t capval, int m, int32_t* rdc, int32_t* binmed, int64_t* binsum,
                            uint32_t* res_hist, BinAccum* acc, hipStream_t stream) {
  // TB bins per tile: as many as fit ~48 KB of values, 4..64, power of two
  int TB = 64;
  while (TB > 4 && (size_t)TB * m * 4 > 48 * 1024) TB >>= 1;
  // LDS histogram range: covers the capped values when the cap is active, 64..256
  int vr = 64;
  while (vr < 256 && vr <= capval) vr <<= 1;
  const size_t tile_pad = ((size_t)TB * m + 3) & ~(size_t)3;
  const size_t lds = tile_pad * 4 + (size_t)vr * kResClasses * 4;
  const int64_t ntiles = (ncompact + (int64_t)TB * m - 1) / ((int64_t)TB * m);
  int grid = (int)(ntiles < 256 * 3 ? (ntiles < 1 ? 1 : ntiles) : 256 * 3);
  hipLaunchKernelGGL(k_cap_compact_bin, dim3(grid), dim3(kThreads), lds, stream, src, n, cbreak, cum, nreg, ncompact, capval,
                     m, TB, vr, rdc, binmed, binsum, res_hist, acc);
}

}  // namespace rsik
