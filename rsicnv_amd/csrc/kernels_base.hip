// kernels_base.hip -- per-base streaming kernels (gfx950): FASTA classification, GC table,
// GC rescale, cap + N-compaction + bin reduction.  All are HBM-bound integer/byte work
// (DESIGN.md section 4): coalesced 16-byte loads, bit-packed GC masks staged through LDS,
// LDS-privatised histograms with bank-spreading replicas, persistent grids sized to the chip.
//
// Built with -ffp-contract=off: the only floating-point here is the GC rescale, which must round
// exactly like the reference's x86-64 SSE2 build.
#include <stdio.h>
#include <algorithm>
#include "kernels.h"
#include "device_util.h"
#include "per_base_device.h"

namespace rsik {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxGrid = 256 * 8;          // 256 CUs x 8 resident workgroups
constexpr int kGcWords = kTileBases / 64;  // 64 words per tile
constexpr int kGcLeft = 4;                 // margin words left of the tile (256 bits >= 201)
constexpr int kGcRight = 2;                // margin words right of the tile (128 bits >= 101)
constexpr int kGcLds = kGcLeft + kGcWords + kGcRight;   // 70

constexpr int kK4GcWords = 128;             // staged GC words of a K4j tile: (6656 + 200) / 64 + 3 <= 111
constexpr int kK3Width = 512, kK3PhaseShift = 4, kK3Phases = 1 << kK3PhaseShift;   // K3': its LDS value histogram is [512 values][16 lane phases]
constexpr int kValLds = 256;               // K3 / K3': values below this are counted in LDS, [value][32 lane phases]


// ------------------------------------------------------------------------------------------
// K1  fasta_classify: one thread per 16 bytes; 4 neighbouring lanes assemble one 64-bit word.
__global__ __launch_bounds__(kThreads) void k_fasta_classify(const uint8_t* __restrict__ fasta, int64_t n,
                                                             uint64_t* __restrict__ gcbits,
                                                             uint64_t* __restrict__ nbits, int64_t nwords, FillList fill) {
  fill_ranges(fill);   // accumulators of the kernels that follow in the stream (first kernel of a chromosome's chain)
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
    vg |= dpp_xor1_u64(vg); vg |= dpp_xor2_u64(vg);   // (quad permutes: one VALU instruction per half instead of a trip through the LDS pipeline)
    vn |= dpp_xor1_u64(vn); vn |= dpp_xor2_u64(vn);
    if (sub == 0) { gcbits[g >> 2] = vg; nbits[g >> 2] = vn; }
  }
}

// K1b n_transitions: run starts and (exclusive) ends of the N mask.  With `pp` the launch's last workgroup also turns the list
// into what K4j needs -- the padded, merged regions of get_noseq_regions (loaddata.cpp:243-273) as compacted break points
// and removed lengths (cbreak / cum) -- so that K4j can be queued behind K2j without the host in between; the host builds
// the same regions from the same list for its own stages.  Lists of more than kRegSortMax entries are left to the host.
constexpr int kRegSortMax = 1024;
// run starts / ends in the words this workgroup takes (a grid-stride loop over the mask words): appended unordered to list[]
__device__ inline void n_transitions_scan(const uint64_t* __restrict__ nbits, int64_t nwords, uint64_t* __restrict__ list,
                                          uint32_t* __restrict__ count, uint32_t cap) {
  // four of a thread's strided words per trip, their eight loads in flight together (256 workgroups: sixty dependent trips per
  // thread on a 250 Mb chromosome otherwise)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t w0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w0 < nwords; w0 += 4 * stride) {
    uint64_t curv[4], prevv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t w = w0 + e * stride;
      curv[e] = w < nwords ? nbits[w] : 0;
      prevv[e] = (w < nwords && w > 0) ? nbits[w - 1] : 0;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t w = w0 + e * stride;
      if (w >= nwords) break;
      const uint64_t cur = curv[e];
      const uint64_t prev_top = prevv[e] >> 63;
      const uint64_t shifted = (cur << 1) | prev_top;
      uint64_t starts = cur & ~shifted, ends = ~cur & shifted;
      while (starts) {
        const int b = __ffsll((long long)starts) - 1;
        starts &= starts - 1;
        const uint32_t k = atomicAdd(count, 1u);
        if (k < cap) st_cg(reinterpret_cast<unsigned long long*>(&list[k]), ((unsigned long long)(w * 64 + b) << 1));
      }
      while (ends) {
        const int b = __ffsll((long long)ends) - 1;
        ends &= ends - 1;
        const uint32_t k = atomicAdd(count, 1u);
        if (k < cap) st_cg(reinterpret_cast<unsigned long long*>(&list[k]), ((unsigned long long)(w * 64 + b) << 1) | 1ull);
      }
    }
  }
}
// the whole list is there (the caller is the launch's last workgroup, all of its threads): sorted, padded by dx, merged -> cbreak / cum
// / pp's region fields.  s_e, s_s: kRegSortMax words of LDS each.
__device__ inline void n_regions_build(const uint64_t* __restrict__ list, const uint32_t* __restrict__ count, int64_t n, int dx,
                                       PhaseParams* __restrict__ pp, int64_t* __restrict__ cbreak, int64_t* __restrict__ cum,
                                       unsigned long long* s_e, unsigned long long* s_s) {
  const unsigned int cnt = ld_cg(count);
  if (cnt > (unsigned int)kRegSortMax || (cnt & 1u)) { if (threadIdx.x == 0) { pp->regions_ok = 0; pp->nreg = 0; pp->ncompact = n; } return; }
  for (unsigned int k = threadIdx.x; k < cnt; k += blockDim.x) s_e[k] = ld_cg(reinterpret_cast<const unsigned long long*>(&list[k]));
  __syncthreads();
  for (unsigned int k = threadIdx.x; k < cnt; k += blockDim.x) {   // rank sort: the entries are distinct
    const unsigned long long e = s_e[k];
    unsigned int rank = 0;
    for (unsigned int j = 0; j < cnt; ++j) rank += s_e[j] < e;
    s_s[rank] = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int nreg = 0, ok = 1;
    int64_t removed = 0, cur_s = 0, cur_e = -2;   // the open (padded) region
    bool open = false;
    for (unsigned int i = 0; i < cnt; i += 2) {
      const unsigned long long a = s_s[i], b = s_s[i + 1];
      if ((a & 1ull) || !(b & 1ull)) { ok = 0; break; }            // a start, then its end
      int64_t gs = (int64_t)(a >> 1) - dx, ge = (int64_t)(b >> 1) - 1 + dx;   // inclusive run, padded by dx each side
      gs = gs < 0 ? 0 : gs; ge = ge > n - 1 ? n - 1 : ge;
      if (open && gs <= cur_e + 1) { cur_e = ge > cur_e ? ge : cur_e; continue; }
      if (open) { cbreak[nreg] = cur_s - removed; removed += cur_e - cur_s + 1; ++nreg; cum[nreg] = removed; }
      cur_s = gs; cur_e = ge; open = true;
    }
    cum[0] = 0;
    if (open) { cbreak[nreg] = cur_s - removed; removed += cur_e - cur_s + 1; ++nreg; cum[nreg] = removed; }
    pp->nreg = nreg; pp->ncompact = n - removed; pp->regions_ok = ok;
  }
}
__global__ __launch_bounds__(kThreads) void k_n_transitions(const uint64_t* __restrict__ nbits, int64_t nwords,
                                                            uint64_t* __restrict__ list, uint32_t* __restrict__ count,
                                                            uint32_t cap, int64_t n, int dx, PhaseParams* __restrict__ pp,
                                                            int64_t* __restrict__ cbreak, int64_t* __restrict__ cum,
                                                            unsigned int* __restrict__ counter) {
  n_transitions_scan(nbits, nwords, list, count, cap);
  if (!pp) return;
  if (!last_block_done(counter)) return;
  __shared__ unsigned long long s_e[kRegSortMax], s_s[kRegSortMax];
  n_regions_build(list, count, n, dx, pp, cbreak, cum, s_e, s_s);
}

// ------------------------------------------------------------------------------------------
// GC window counts.  The tile's GC words (with margins) are staged in LDS together with an
// exclusive prefix of their popcounts; rank(x) = #GC in [first staged bit, x).
struct GcTile {
  uint64_t word[kGcLds];
  uint32_t pre[kGcLds + 1];
};

// The tile's GC words are requested one tile ahead (by wave 0, two words per lane) and committed
// to LDS, with their popcount prefix, when the tile is processed.
struct GcRegs { uint64_t a, b; };

__device__ inline void gc_tile_request(GcRegs& g, const uint64_t* __restrict__ gcbits, int64_t nwords, int64_t tile_word0) {
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    const int64_t gw = tile_word0 - kGcLeft + l, gw2 = gw + 64;
    // out-of-range words read word 0 and are zeroed at commit (keeps the loads branch-free)
    g.a = gcbits[(gw >= 0 && gw < nwords) ? gw : 0];
    g.b = gcbits[(l < kGcLds - 64 && gw2 >= 0 && gw2 < nwords) ? gw2 : 0];
  }
}

__device__ inline void gc_tile_commit(GcTile& t, const GcRegs& g, int64_t nwords, int64_t tile_word0) {
  // wave 0 writes and scans; callers __syncthreads() afterwards
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    const int64_t gw = tile_word0 - kGcLeft + l, gw2 = gw + 64;
    const uint64_t w = (gw >= 0 && gw < nwords) ? g.a : 0;
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
    const uint64_t w2 = (l < kGcLds - 64 && gw2 >= 0 && gw2 < nwords) ? g.b : 0;
    if (l < kGcLds - 64) t.word[64 + l] = w2;
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
// Shared skeleton of the two GC streaming kernels (K2, K3).  Per tile of 4096 bases:
//   1. the tile's depth is requested with four coalesced 16-byte loads per thread, one tile ahead
//      of its use (registers double-buffered), so a workgroup always has 16 KB in flight;
//   2. meanwhile the window GC count of every base of the tile is produced cooperatively into LDS
//      as one byte per base: each thread takes 16 consecutive bases, two rank queries for the
//      first and one leaving/entering bit pair for each of the others;
//   3. each thread then consumes its four quads with one 4-byte LDS read of counts per quad.
struct TileRegs { int4 q[kTileBases / (4 * kThreads)]; };

__device__ inline void tile_request(TileRegs& r, const int32_t* __restrict__ depth, int64_t base, int64_t n) {
  // Branch-free: quads that are not completely inside the array read quad 0 instead (ignored later),
  // so the four loads issue back to back with nothing waiting in between.
#pragma unroll
  for (int k = 0; k < kTileBases / (4 * kThreads); ++k) {
    const int64_t q = base + 4 * (int64_t)(k * kThreads + threadIdx.x);
    const int64_t qs = (q + 4 <= n) ? q : 0;
    r.q[k] = *reinterpret_cast<const int4*>(depth + qs);
  }
}

// The ragged last quad of the chromosome (n % 4 != 0) is left to the tail kernels
// (k_gc_hist_tail, k_gc_tail_fixup): the streaming kernels only consume whole quads, so nothing in
// their loop body waits on a memory operation other than the tile that was requested one trip ago.

// 16 mask bits starting at staged bit `rel` (rel + 16 stays inside the staged words)
__device__ inline uint32_t gc_field16(const GcTile& t, uint32_t rel) {
  const uint32_t k = rel >> 6, b = rel & 63;
  uint64_t v = t.word[k] >> b;
  if (b > 48) v |= t.word[k + 1] << (64 - b);
  return (uint32_t)v & 0xffffu;
}

// Window counts of the tile's bases into s_g (one byte each).  Thread t covers bases 16t..16t+15.
__device__ inline void tile_window_counts(const GcTile& gt, unsigned char* s_g, int64_t base, int64_t n, int64_t first_bit) {
  const int64_t i0 = base + 16 * (int64_t)threadIdx.x;
  uint32_t c[16];
  const bool interior = base >= 101 && base + kTileBases - 1 <= n - 102;   // no edge clamping anywhere in the tile
  if (interior) {
    const uint32_t rel = (uint32_t)(i0 - 100 - first_bit);
    uint32_t cnt = gc_rank(gt, rel + 201) - gc_rank(gt, rel);
    const uint32_t leave = gc_field16(gt, rel), enter = gc_field16(gt, rel + 201);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      c[j] = cnt;
      cnt = cnt - ((leave >> j) & 1u) + ((enter >> j) & 1u);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = (i0 + j < n) ? (uint32_t)gc_window(gt, i0 + j, n, first_bit) : 0u;
  }
  uint4 packed;
  packed.x = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
  packed.y = c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24);
  packed.z = c[8] | (c[9] << 8) | (c[10] << 16) | (c[11] << 24);
  packed.w = c[12] | (c[13] << 8) | (c[14] << 16) | (c[15] << 24);
  *reinterpret_cast<uint4*>(s_g + 16 * threadIdx.x) = packed;
}

// ------------------------------------------------------------------------------------------
// K2  gc_hist, wave-autonomous.  Each wave owns sub-tiles of 1024 bases (16 consecutive bases per
// lane) and never meets a workgroup barrier inside its loop: its GC words (16 + margins) go to a
// per-wave LDS slot, a lane derives its 16 window counts from two rank queries plus one
// leaving/entering bit pair per base, and adds its depths straight from registers into the
// workgroup's LDS accumulators.  The 64 bytes a lane needs are four 16-byte loads, requested one
// sub-tile ahead (two register sets used alternately).
//
// PACKED: one 64-bit LDS atomic adds (1 << 40) + depth, i.e. count and sum together; valid while
// every depth is below 2^21 and a workgroup sees at most 2^18 bases (sum field < 2^40, count field
// < 2^24).  Larger depths raise flag bit 1 and the host re-runs the two-atomic form.
constexpr int kGcRep = 16;  // LDS replicas per GC level, selected by lane & 15
constexpr int kSubBases = 1024;                       // bases per wave trip
constexpr int kSubWords = kSubBases / 64;             // 16
constexpr int kSubLds = kGcLeft + kSubWords + kGcRight;   // 22 staged words per wave
constexpr int kGcMaxSubPerWg = 256;                   // 2^18 bases per workgroup at most (packed fields)

struct WaveGc {            // per-wave LDS slot
  uint64_t word[kSubLds + 2];
};

// GC bases among the 201 starting at bit `lo` of the mask (0 <= lo, lo + 201 <= number of bits): popcounts of the four or
// five words the window touches -- the tails of K2 and K3 used to walk it bit by bit, 201 dependent loads per cell.
__device__ inline int gc_window_count(const uint64_t* __restrict__ gcbits, int64_t lo) {
  const int64_t hi = lo + 201;   // exclusive
  int g = 0;
  for (int64_t w = lo >> 6; w <= (hi - 1) >> 6; ++w) {
    uint64_t x = gcbits[w];
    const int64_t b0 = w << 6;
    if (lo > b0) x &= ~0ull << (lo - b0);
    if (hi < b0 + 64) x &= (1ull << (hi - b0)) - 1;
    g += __popcll(x);
  }
  return g;
}
// GC bases among the 201 from staged bit `rel` on: popcounts of the five words the window can touch, all five reads
// independent.  (The first version kept a popcount prefix per wave and took two rank queries; building the prefix -- five
// dependent cross-lane steps per sub-tile -- was on the critical path of every trip of K2 and K3'.)
__device__ inline uint32_t wgc_window(const WaveGc& t, uint32_t rel) {
  const uint32_t k = rel >> 6, b = rel & 63;
  const uint64_t w0 = t.word[k], w1 = t.word[k + 1], w2 = t.word[k + 2], w3 = t.word[k + 3], w4 = t.word[k + 4];
  const uint32_t rem = 9 + b;                                   // bits of the window behind the first three words: 9 .. 72
  const uint64_t m3 = rem >= 64 ? ~0ull : ((1ull << rem) - 1);
  const uint64_t m4 = rem > 64 ? ((1ull << (rem - 64)) - 1) : 0ull;
  return (uint32_t)(__popcll(w0 >> b) + __popcll(w1) + __popcll(w2) + __popcll(w3 & m3) + __popcll(w4 & m4));
}
__device__ inline uint32_t wgc_field16(const WaveGc& t, uint32_t rel) {
  const uint32_t k = rel >> 6, b = rel & 63;
  uint64_t v = t.word[k] >> b;
  if (b > 48) v |= t.word[k + 1] << (64 - b);
  return (uint32_t)v & 0xffffu;
}


struct SubRegs { int4 q[4]; uint64_t gw; };

__device__ inline void sub_request(SubRegs& r, const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits,
                                   int64_t nwords, int64_t base, int64_t n, int lane) {
  const int64_t i0 = base + 16 * (int64_t)lane;
#pragma unroll
  for (int k = 0; k < 4; ++k) {   // branch-free: quads not wholly inside the array read quad 0 (ignored later)
    const int64_t q = i0 + 4 * k;
    r.q[k] = *reinterpret_cast<const int4*>(depth + (q + 4 <= n ? q : 0));
  }
  const int64_t w = base / 64 - kGcLeft + lane;
  r.gw = gcbits[(lane < kSubLds && w >= 0 && w < nwords) ? w : 0];
}

// Workgroup results leave as one plain, coalesced slab per workgroup (kGcSlab 64-bit words:
// sum[202], cnt[202], possum, poscnt, flags) and are folded by k_gc_hist_reduce.  Thousands of
// workgroups adding into the same ~400 global words with atomics serialise on those words and
// cost more than the whole streaming pass.
constexpr int kGcSlab = 2 * kGcLevels + 4;

template <bool PACKED>
__global__ __launch_bounds__(kThreads, 3) void k_gc_hist(const int32_t* __restrict__ depth,
                                                      const uint64_t* __restrict__ gcbits, int64_t n, int64_t nwords,
                                                      unsigned long long* __restrict__ slabs, unsigned long long* __restrict__ gsum,
                                                      int per_group, unsigned int* __restrict__ counters,
                                                      GcAccum* __restrict__ acc, double* __restrict__ table, uint8_t* __restrict__ d8) {
  __shared__ WaveGc s_gc[kThreads / 64];
  __shared__ unsigned long long s_tail[4 * (kThreads / 64)];
  __shared__ unsigned long long s_sum[kGcLevels * kGcRep];
  __shared__ unsigned int s_cnt[PACKED ? 1 : kGcLevels * kGcRep];
  for (int e = threadIdx.x; e < kGcLevels * kGcRep; e += kThreads) { s_sum[e] = 0; if (!PACKED) s_cnt[e] = 0; }
  __syncthreads();
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  WaveGc& G = s_gc[wave];
  const int rep = lane & (kGcRep - 1);
  unsigned long long possum = 0, poscnt = 0;
  unsigned int escapes = 0;   // per lane: far below 2^32
  int vmax = 0, vmin = 0;
  const int64_t nsub = (n + kSubBases - 1) / kSubBases;
  const int64_t stride = (int64_t)gridDim.x * (kThreads / 64);
  // the byte copy of the depth (kByteEscape = "255 or more: look at the int32 array"), what K3 and K4 stream instead
  auto sat8 = [](int v) -> uint32_t { return v < 0 ? 0u : (v >= kByteEscape ? (uint32_t)kByteEscape : (uint32_t)v); };

  auto trip = [&](const SubRegs& cur, SubRegs& nxt, int64_t sub) {
    const int64_t base = sub * kSubBases;
    const int64_t first_bit = base - kGcLeft * 64;
    // ---- commit this sub-tile's GC words to the wave's LDS slot ----
    {
      const int64_t w = base / 64 - kGcLeft + lane;
      const uint64_t word = (lane < kSubLds && w >= 0 && w < nwords) ? cur.gw : 0;
      if (lane < kSubLds + 1) G.word[lane] = word;
    }
    if (sub + stride < nsub) sub_request(nxt, depth, gcbits, nwords, (sub + stride) * kSubBases, n, lane);
    __builtin_amdgcn_wave_barrier();
    // ---- the lane's 16 window counts ----
    const int64_t i0 = base + 16 * (int64_t)lane;
    uint32_t cnt0, leave, enter;
    const bool interior = base >= 101 && base + kSubBases - 1 <= n - 102;   // no edge clamping in this sub-tile
    if (interior) {
      const uint32_t rel = (uint32_t)(i0 - 100 - first_bit);
      cnt0 = wgc_window(G, rel);
      leave = wgc_field16(G, rel); enter = wgc_field16(G, rel + 201);
    } else {
      cnt0 = leave = enter = 0;
    }
    uint32_t cnt = cnt0;
    unsigned long long pend = 0; uint32_t pend_g = 0xffffffffu; unsigned int pend_c = 0;
    auto flush = [&]() {
      if (pend_g == 0xffffffffu) return;
      if (PACKED) atomicAdd(&s_sum[pend_g * kGcRep + rep], pend);
      else { atomicAdd(&s_sum[pend_g * kGcRep + rep], pend); atomicAdd(&s_cnt[pend_g * kGcRep + rep], pend_c); }
      pend_g = 0xffffffffu;
    };
    auto add = [&](int val, uint32_t g) {
      if (g != pend_g) { flush(); pend_g = g; pend = 0; pend_c = 0; }
      pend += PACKED ? (1ull << 40) + (unsigned long long)(unsigned int)val : (unsigned long long)(long long)val;
      pend_c += 1;
      const int pos = val > 0 ? val : 0;
      possum += (unsigned long long)pos; poscnt += val > 0;
      vmax = val > vmax ? val : vmax; vmin = val < vmin ? val : vmin;
    };
    if (interior) {   // every sub-tile but the first and the last one or two: straight-line code
      auto one = [&](int val, int j) {
        add(val, cnt);
        cnt = cnt - ((leave >> j) & 1u) + ((enter >> j) & 1u);
      };
#pragma unroll
      for (int k = 0; k < 4; ++k) { one(cur.q[k].x, 4 * k); one(cur.q[k].y, 4 * k + 1); one(cur.q[k].z, 4 * k + 2); one(cur.q[k].w, 4 * k + 3); }
      if (d8) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          w[k] = sat8(cur.q[k].x) | (sat8(cur.q[k].y) << 8) | (sat8(cur.q[k].z) << 16) | (sat8(cur.q[k].w) << 24);
          escapes += (cur.q[k].x >= kByteEscape) + (cur.q[k].y >= kByteEscape) + (cur.q[k].z >= kByteEscape) + (cur.q[k].w >= kByteEscape);
        }
        *reinterpret_cast<uint4*>(d8 + i0) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    } else {          // edge sub-tiles: the reference's clamped windows (App. A Q1), whole quads only
      for (int j = 0; j < 16; ++j) {
        const int64_t i = i0 + j;
        if ((i & ~(int64_t)3) + 4 > n) break;
        int64_t lo = i - 100;
        if (lo < 0) lo = 0;
        if (lo > n - 202) lo = n - 202;
        const uint32_t rel = (uint32_t)(lo - first_bit);
        const int val = depth[i];
        add(val, wgc_window(G, rel));
        if (d8) { d8[i] = (uint8_t)sat8(val); escapes += val >= kByteEscape; }
      }
    }
    flush();
    __builtin_amdgcn_wave_barrier();   // the slot is rewritten by the next trip
  };

  SubRegs ra, rb;
  int64_t sub = (int64_t)blockIdx.x * (kThreads / 64) + wave;
  if (sub < nsub) sub_request(ra, depth, gcbits, nwords, sub * kSubBases, n, lane);
  while (sub < nsub) {
    trip(ra, rb, sub);
    sub += stride;
    if (sub >= nsub) break;
    trip(rb, ra, sub);
    sub += stride;
  }
  unsigned int neg = vmin < 0 ? 1u : 0u;
  if (PACKED && vmax >= (1 << 21)) neg |= 2;   // packed fields could overflow: results of this launch are discarded
  for (int d = 32; d >= 1; d >>= 1) {
    possum += __shfl_xor(possum, d);
    poscnt += __shfl_xor(poscnt, d);
    escapes += __shfl_xor(escapes, d);
    neg |= __shfl_xor(neg, d);
  }
  if (lane == 0) { s_tail[4 * wave] = possum; s_tail[4 * wave + 1] = poscnt; s_tail[4 * wave + 2] = neg; s_tail[4 * wave + 3] = escapes; }
  __syncthreads();
  unsigned long long* slab = slabs + (size_t)blockIdx.x * kGcSlab;
  for (int g = threadIdx.x; g < kGcLevels; g += kThreads) {
    unsigned long long s2 = 0, c = 0;
    for (int r = 0; r < kGcRep; ++r) {
      const unsigned long long w = s_sum[g * kGcRep + r];
      if (PACKED) { s2 += w & ((1ull << 40) - 1); c += w >> 40; }
      else { s2 += w; c += s_cnt[g * kGcRep + r]; }
    }
    st_cg(&slab[g], s2); st_cg(&slab[kGcLevels + g], c);
  }
  if (threadIdx.x == 0) {
    unsigned long long ps = 0, pc = 0, fl = 0, esc = 0;
    for (int w = 0; w < kThreads / 64; ++w) { ps += s_tail[4 * w]; pc += s_tail[4 * w + 1]; fl |= s_tail[4 * w + 2]; esc += s_tail[4 * w + 3]; }
    // the two flags as additive fields (number of workgroups that raised them), so that the fold can sum the slab
    st_cg(&slab[2 * kGcLevels], ps); st_cg(&slab[2 * kGcLevels + 1], pc); st_cg(&slab[2 * kGcLevels + 2], (fl & 1ull) | ((fl >> 1) << 32)); st_cg(&slab[2 * kGcLevels + 3], esc);
  }
  // ---- the last workgroup to finish folds the slabs (device_util.h), adds the ragged last n % 4 bases, which lie in the
  // stale-window zone i >= n-101 whose count is that of [n-202, n-2] (App. A Q1), and builds the GC table
  // (gccontent.cpp:109-112, 141-145): level means, the mean of the positive depths where a level is empty or below 1;
  // table[kGcLevels] = that mean.  No fold launch, no table launch, no host round trip between K2 and K3. ----
  unsigned long long* total = s_sum;
  if (!fold_slabs(slabs, gsum, total, kGcSlab, per_group, counters)) return;
  if (threadIdx.x == 0 && (n & 3) != 0) {
    int g = 0;
    g = gc_window_count(gcbits, n - 202);   // bits n-202 .. n-2
    for (int64_t i = n & ~(int64_t)3; i < n; ++i) {
      const int v = depth[i];
      total[g] += (unsigned long long)(long long)v;
      total[kGcLevels + g] += 1ull;
      if (v > 0) { total[2 * kGcLevels] += (unsigned long long)v; total[2 * kGcLevels + 1] += 1ull; }
      if (v < 0) total[2 * kGcLevels + 2] |= 1ull;
      if (d8) { d8[i] = (uint8_t)sat8(v); total[2 * kGcLevels + 3] += v >= kByteEscape; }
    }
  }
  __syncthreads();
  const unsigned long long ps = total[2 * kGcLevels], pc = total[2 * kGcLevels + 1], fl = total[2 * kGcLevels + 2];
  double rdmean = (double)ps;
  if (pc > 0) rdmean /= (double)pc;
  for (int g = threadIdx.x; g < kGcLevels; g += kThreads) {
    const unsigned long long sg = total[g], cg = total[kGcLevels + g];
    acc->sum[g] = sg; acc->cnt[g] = cg;
    double t = cg > 0 ? (double)sg / (double)cg : rdmean;
    if (t < 1) t = rdmean;
    table[g] = t;
  }
  if (threadIdx.x == 0) {
    acc->possum = ps; acc->poscnt = pc;
    acc->negatives = ((fl & 0xffffffffull) ? 1u : 0u) | ((fl >> 32) ? 2u : 0u);
    const unsigned long long esc = total[2 * kGcLevels + 3];
    acc->escapes = esc > 0xffffffffull ? 0xffffffffu : (unsigned int)esc;   // bases whose byte copy says "look at the int32 array"
    table[kGcLevels] = rdmean;
  }
}

// Tail of the 20-slice write-back (gccontent.cpp:156-175; App. A Q2/Q3) and the ragged last quad
// that the streaming kernel leaves out.  One wave (the last workgroup of K3).  adjust = 0: only the ragged quad's values are
// added to the histogram (the -NOGC path has no slices).
// The cells' old and new values go through `hist_add(value)` / `hist_sub(value)`: the global histogram (gc_tail_fixup below) or
// a workgroup's LDS copy of it (K2j's tail).
template <class Add, class Sub>
__device__ inline void gc_tail_fixup_with(const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits, int64_t n,
                                          const double* table /* [kGcLevels] + rdmean */, int adjust, int32_t* __restrict__ out,
                                          Add hist_add, Sub hist_sub) {
  // called by wave 0 of one workgroup; lane k handles tail cell k (r <= 19)
  const double rdmean = adjust ? table[kGcLevels] : 0.0;
  const int lane = threadIdx.x;
  const int64_t ragged = n & ~(int64_t)3;   // first base not consumed by the streaming kernel
  // what the streaming loop wrote to out[i]: recomputed here, because a plain load of another workgroup's plain store
  // of the same launch may be stale (device_util.h); window rule of App. A Q1
  auto streamed = [&](int64_t i) {
    int64_t lo = i - 100;
    if (lo < 0) lo = 0;
    if (lo > n - 202) lo = n - 202;
    int g = 0;
    g = gc_window_count(gcbits, lo);
    return (int)((double)depth[i] * rdmean / table[g] + 0.5);
  };
  if (!adjust) {
    if (ragged + lane < n) hist_add(depth[ragged + lane]);
    return;
  }
  const int64_t S = n / 20, r = n - 20 * S;   // r >= n % 4, so the ragged bases are among the last r
  if (r == 0) return;
  // fresh edge window [n-201, n-1]: 201 mask bits counted with four ballots
  int gtail = 0;
  for (int c = 0; c < 4; ++c) {
    const int64_t i = n - 201 + 64 * c + lane;
    const bool bit = i < n && ((gcbits[i >> 6] >> (i & 63)) & 1);
    gtail += __popcll(__ballot(bit));
  }
  // the two groups of cells are disjoint (n-201+k < 20S), and each lane owns one cell of each
  if (r >= 2 && lane < r) {
    const int nv = (int)((double)depth[20 * S + lane] * rdmean / table[gtail] + 0.5);
    const int64_t idx = n - 201 + lane;
    hist_sub(streamed(idx)); hist_add(nv);
    if (out) out[idx] = nv;
  }
  if (lane < r) {   // the last r bases keep their unadjusted depth
    const int64_t idx = 20 * S + lane;
    if (idx < ragged) hist_sub(streamed(idx));
    hist_add(depth[idx]);
    if (out) out[idx] = depth[idx];
  }
}
__device__ inline void gc_tail_fixup(const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits, int64_t n,
                                     const double* __restrict__ table /* [kGcLevels] + rdmean */, int adjust, int32_t* __restrict__ out,
                                     uint32_t* __restrict__ ghist, ValueHistAux* __restrict__ aux) {
  auto hist_add = [&](int to) {
    if (!ghist) return;
    if (to >= 0 && to < kHistValues) { atomicAdd(&ghist[to], 1u); if (to >= kValLds) atomicMax(&aux->vmax, (unsigned int)to); }
    else if (to >= kHistValues) { atomicAdd(&aux->big, 1ull); atomicMax(&aux->vmax, (unsigned int)to); }
    else atomicOr(&aux->negatives, 1u);
  };
  auto hist_sub = [&](int from) {
    if (!ghist) return;
    if (from >= 0 && from < kHistValues) atomicSub(&ghist[from], 1u);
    else if (from >= kHistValues) atomicAdd(&aux->big, (unsigned long long)-1ll);
  };
  gc_tail_fixup_with(depth, gcbits, n, table, adjust, out, hist_add, hist_sub);
}

// Median walk of partition_stat_tp (wufunctions.cpp:398-420, dy = 1) over hist[kHistValues] for `total` values by one
// workgroup of NT threads (kHistValues / NT consecutive counters each); the counters were written by other workgroups.
// `range`: counters at and beyond it are known to be zero (the kernels keep the largest value that went past the LDS range
// in ValueHistAux::vmax), a multiple of NT.
template <int NT>
__device__ inline void value_median_block(const uint32_t* __restrict__ hist, unsigned long long total, ValueMedian* __restrict__ out, int range) {
  __shared__ unsigned long long s_w[NT / 64];
  __shared__ int s_lo[NT / 64], s_hi[NT / 64], s_med;
  const int kPer = (range + NT - 1) / NT;   // hist has kHistValues counters: a stretch may run past `range` (zeros there), never past the array
  const int v0 = threadIdx.x * kPer;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long local = 0;
  int lo = 0x7fffffff, hi = -1;
#pragma unroll 8
  for (int i = 0; i < kPer; ++i) {
    const uint32_t c = v0 + i < kHistValues ? ld_cg(hist + v0 + i) : 0u;
    local += c;
    if (c) { lo = v0 + i < lo ? v0 + i : lo; hi = v0 + i; }
  }
  unsigned long long incl = local;
  for (int d = 1; d < 64; d <<= 1) { const unsigned long long up = __shfl_up(incl, d); if (lane >= d) incl += up; }
  int wlo = lo, whi = hi;
  for (int d = 32; d >= 1; d >>= 1) { const int a = __shfl_xor(wlo, d), b = __shfl_xor(whi, d); wlo = a < wlo ? a : wlo; whi = b > whi ? b : whi; }
  if (lane == 63) s_w[wave] = incl;
  if (lane == 0) { s_lo[wave] = wlo; s_hi[wave] = whi; }
  if (threadIdx.x == 0) s_med = -1;
  __syncthreads();
  unsigned long long base = 0, all = 0;
  int glo = 0x7fffffff, ghi = -1;
  for (int w = 0; w < NT / 64; ++w) { if (w < wave) base += s_w[w]; all += s_w[w]; glo = s_lo[w] < glo ? s_lo[w] : glo; ghi = s_hi[w] > ghi ? s_hi[w] : ghi; }
  const unsigned long long r2 = total / 2;
  unsigned long long seen = base + incl - local;
  if (local != 0 && seen < r2 && seen + local >= r2) {   // the walk's bucket lies in this thread's stretch
    for (int i = 0; i < kPer && v0 + i < kHistValues; ++i) {
      const unsigned long long upto = seen + ld_cg(hist + v0 + i);
      if (seen < r2 && upto >= r2) s_med = v0 + i;
      seen = upto;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {   // agent-scope stores: the same workgroup reads the record back through ld_cg for the export
    st_cg(&out->inrange, all);
    st_cg(reinterpret_cast<unsigned int*>(&out->lo), (unsigned int)glo); st_cg(reinterpret_cast<unsigned int*>(&out->hi), (unsigned int)ghi);
    st_cg(reinterpret_cast<unsigned int*>(&out->med), (unsigned int)s_med); st_cg(reinterpret_cast<unsigned int*>(&out->pad), 0u);
  }
}


// The LDS value histogram of K3 holds 8192 counters: [256 values][32 lane phases] from 0 for ordinary coverage; for a mean
// depth of 160 and more [1024 values][8 lane phases] from mean - 384 (hist_window_base below): deep distributions are wide,
// so lanes seldom meet on one value, and every value outside the window is a global atomic on one of a few hundred words --
// with 4 % of the bases outside, that alone took 60 times the streaming pass.  Every workgroup derives the same window
// from the GC table's mean depth.
constexpr int kDeepWidth = 1024, kDeepPhaseShift = 3;
__device__ inline int hist_window_base_dev(double center, int width) {
  if (!(center >= 160.0)) return 0;
  const double b = center - (double)(3 * width / 8);
  return b > (double)(kHistValues - width) ? kHistValues - width : (b < 0.0 ? 0 : (int)b);
}
// rare path (values outside the LDS range): kept out of line so the unrolled callers stay small
__device__ __attribute__((noinline)) void value_hist_add(unsigned int* s_hist, uint32_t* __restrict__ ghist, ValueHistAux* aux, int v,
                                      int phase, int vb = 0 /* first value of the LDS window */, int width = kValLds, int phsh = 5) {
  if (v >= vb && v - vb < width) atomicAdd(&s_hist[((v - vb) << phsh) + phase], 1u);
  else if (v >= 0 && v < kHistValues) atomicAdd(&ghist[v], 1u);   // the caller keeps the largest such value (lane register -> ValueHistAux::vmax)
  else if (v < 0) atomicOr(&aux->negatives, 1u);
  else atomicAdd(&aux->big, 1ull);
}

// The largest value a workgroup's lanes counted outside the LDS range -> ValueHistAux::vmax: one atomic per wave that has
// one (a same-address atomic per VALUE serialises: a chromosome with ten thousand such values lost a millisecond to it).
__device__ inline void publish_hist_hi(int lane_hi, ValueHistAux* __restrict__ aux) {
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(lane_hi, d); lane_hi = o > lane_hi ? o : lane_hi; }
  if (lane_id() == 0 && lane_hi >= kValLds) atomicMax(&aux->vmax, (unsigned int)lane_hi);
}

// What the last workgroup of a value-histogram kernel (K3, K3') does once every workgroup's LDS histogram s_hist
// ([kValLds][32 lane phases]) is complete: per-workgroup slab, fold (device_util.h), tail quirks of the 20-slice
// write-back, the walk to the median apply_cap needs (loaddata.cpp:233), and the chromosome's header (GC accumulators,
// counters, median, the first N-run entries) into mapped host memory: what used to be three launches and a device -> host
// copy behind the kernel.  All threads of every workgroup call it.
template <bool ADJUST, bool WITH_CAP = false>
__device__ inline void value_hist_finish(unsigned int* s_hist, const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits,
                                         int64_t n, const double* __restrict__ table, int32_t* __restrict__ out,
                                         uint32_t* __restrict__ ghist, ValueHistAux* __restrict__ aux,
                                         unsigned int* __restrict__ hist_slabs, unsigned int* __restrict__ gsum, int per_group,
                                         unsigned int* __restrict__ counters, ValueMedian* __restrict__ vm, const void* head_src,
                                         void* head_dst, unsigned int head_bytes, int vb = 0 /* first value of the LDS window */,
                                         int width = kValLds, int phsh = 5, PhaseParams* __restrict__ pp = nullptr /* the cap for a K4 queued behind this launch */,
                                         double cap_mult = 0.0) {
  const int phases = 1 << phsh;
  unsigned int mine[kDeepWidth / kThreads];   // all reads of the [value][phase] counters come before the first write of a total
#pragma unroll
  for (int k = 0; k < kDeepWidth / kThreads; ++k) {
    const int v = k * kThreads + (int)threadIdx.x;
    unsigned int c = 0;
    if (v < width) for (int p = 0; p < phases; ++p) c += s_hist[(v << phsh) + ((p + v) & (phases - 1))];
    mine[k] = c;
  }
#pragma unroll
  for (int k = 0; k < kDeepWidth / kThreads; ++k) {
    const int v = k * kThreads + (int)threadIdx.x;
    if (v < width) st_cg(&hist_slabs[(size_t)blockIdx.x * width + v], mine[k]);
  }
  unsigned int* total = s_hist;
  if (!fold_slabs(hist_slabs, gsum, total, width, per_group, counters)) return;
  for (int v = threadIdx.x; v < width; v += kThreads) { const unsigned int c = total[v]; if (c) atomicAdd(&ghist[vb + v], c); }
  sync_drained();
  // histogram side of the tail quirks only: out[]'s tail cells were stored by other workgroups of this launch, and a second
  // store from here would race with them (no defined order between XCDs); the launcher rewrites them in a launch of their own
  if (threadIdx.x < 64 && (ADJUST || (n & 3) != 0)) gc_tail_fixup(depth, gcbits, n, table, ADJUST ? 1 : 0, nullptr, ghist, aux);
  sync_drained();
  {
    const unsigned int hi = ld_cg(&aux->vmax);                       // largest value counted outside the LDS range (0: none)
    int range = hi >= (unsigned int)kHistValues ? kHistValues : (int)hi + 1;
    range = range < vb + width ? vb + width : range;
    range = (range + kThreads - 1) / kThreads * kThreads;
    value_median_block<kThreads>(ghist, (unsigned long long)n, vm, range > kHistValues ? kHistValues : range);
  }
  sync_drained();
  if (WITH_CAP && pp && threadIdx.x == 0) {
    // the cap as apply_cap takes it (loaddata.cpp:233-238), as K2j's tail leaves it: for a K4s / K4m queued right behind this launch
    // (-NOGC); the host derives the same number from the header and checks everything else
    const int lo = (int)ld_cg(reinterpret_cast<const unsigned int*>(&vm->lo)), hi = (int)ld_cg(reinterpret_cast<const unsigned int*>(&vm->hi));
    const int med = (int)ld_cg(reinterpret_cast<const unsigned int*>(&vm->med));
    double qm = (double)lo;
    if (lo <= hi && (double)hi - (double)lo >= 1.0 && med >= 0) qm = (double)med;
    const unsigned long long inr = ld_cg(&vm->inrange);
    const bool ok = cap_mult > 1.0 && inr + ld_cg(&aux->big) == (unsigned long long)n && (unsigned long long)n / 2 <= inr && ld_cg(&aux->negatives) == 0u;
    st_cg(reinterpret_cast<unsigned int*>(&pp->capval), (unsigned int)(ok ? (int32_t)(qm * cap_mult) : -1));
  }
  sync_drained();
  export_words(head_dst, head_src, head_bytes);
}

// ------------------------------------------------------------------------------------------
// K2j  gc_joint_hist: K2 with the JOINT histogram H[window GC count][depth byte] instead of a sum and a count per GC count.
// Everything K3' computed per base follows from H without touching the bases again -- the rescaled value is a pure function
// of (depth, GC count): (int)(d * rdmean / table[g] + 0.5), gccontent.cpp:89 -- so the last workgroup of THIS launch builds the
// GC table (row sums of H), the histogram of the rescaled depth (202 x 255 evaluations of that expression instead of n) and
// the cap median from it, and the chromosome is one per-base pass shorter: K1, K2j, K4j.
//
// LDS: 202 x 256 counters of 16 bits, two to a word (103 KB: one workgroup of kJWaves waves per CU, wave-autonomous trips as in
// K2).  A 16-bit counter can wrap; nothing in the loop checks for it.  Instead every workgroup compares the sum of its fields
// with the number of bases it counted: a wrapped low field loses 65536 and carries 1 into its neighbour, a wrapped high field
// loses 65536, so the sum comes out short whenever anything wrapped -- the workgroup then raises a flag and the host sends the
// chromosome through the three-pass chain (K2, K3', K4j).  It takes a cell with more than 6 % of a workgroup's million bases:
// depth 0 has 32-bit counters of its own (s_zero: N runs and uncovered stretches put megabases into one cell), a sequence
// without any GC variation next to a constant depth is what is left.  Depths of 255 and more ("escapes", the byte copy says
// kByteEscape) are summed per GC count for the table (64-bit) and enter the value histogram through a pass of their own
// (k_escape_hist), launched by the host only when the header reports any.
constexpr int kJWaves = 12, kJThreads = 64 * kJWaves;
constexpr int kJRowWords = 128;                                  // one GC level: values 2k (low half) and 2k + 1 (high half) in word k
constexpr int kJPacked = kGcLevels * kJRowWords;                 // 25856 words
constexpr int kJOffZero = kJPacked;                              // [202] bases of depth 0 per level
constexpr int kJOffEscCnt = kJOffZero + kGcLevels;               // [202] escapes per level
constexpr int kJOffFlags = kJOffEscCnt + kGcLevels;              // [8]: workgroups that saw a negative depth, escapes, workgroups whose counters wrapped
constexpr int kJOffEscSum = kJOffFlags + 8;                      // [202] 64-bit sums of the escapes' depths (8-byte aligned)
constexpr int kJSlabWords = kJOffEscSum + 2 * kGcLevels;         // 26672, a multiple of 4
static_assert(kJSlabWords % 4 == 0 && (kJOffEscSum % 2) == 0 && kJPacked % 4 == 0, "slab layout");
// A workgroup's pair counters are mostly zero (a chromosome's window GC counts cover 50 - 80 of the 202 levels, its depths a
// quarter of the 256 values): the slab is written, and folded, in BLOCKS of 32 words (a level's row = 4 blocks of 64 values),
// only the blocks that hold a count.  Behind the slab image: the workgroup's block bitmap (26 words), then words that stay zero
// (the quad at +28 is what a folding lane reads in place of a block its member did not write).
constexpr int kJBlocks = kJPacked / 32;                          // 808
constexpr int kJBitWords = (kJBlocks + 31) / 32;                 // 26
constexpr int kJSlabStride = kJSlabWords + 32;                   // words between two workgroups' slabs
constexpr int kJZeroQuad = kJSlabWords + 28;
static_assert(kJPacked % 32 == 0 && kJBitWords <= 28 && kJSlabStride % 4 == 0, "block bitmap layout");
// the folded totals (global, zero before the launch): H as 32-bit counters, then the slab's tail as it is
constexpr int kJTotH = kGcLevels * 256;
constexpr int kJTotWords = kJTotH + (kJSlabWords - kJPacked);
constexpr int kJRh = 8192;                                       // rescaled values the tail counts in LDS (beyond: global atomics)
constexpr double kFixMaxRatio = 3.99;                            // R = ratio * 2^22 stays below 2^24
constexpr int kJEscPerWg = 63;                                   // escapes a workgroup lists by position (entry 0 of its list: their number)

// Eight 16-byte loads of other workgroups' results (sc1), all in flight at once: the thread's offsets are eight VGPRs, the base
// is wave-uniform.  The tail of K2j walks 200 KB with them; one load and one wait at a time was 50 us of a 200 us kernel.
__device__ inline void ld_cg_x4_batch8(u32x4 (&v)[8], unsigned long long base, const unsigned int (&off)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, %16 sc1\n\t"
      "global_load_dwordx4 %1, %9, %16 sc1\n\t"
      "global_load_dwordx4 %2, %10, %16 sc1\n\t"
      "global_load_dwordx4 %3, %11, %16 sc1\n\t"
      "global_load_dwordx4 %4, %12, %16 sc1\n\t"
      "global_load_dwordx4 %5, %13, %16 sc1\n\t"
      "global_load_dwordx4 %6, %14, %16 sc1\n\t"
      "global_load_dwordx4 %7, %15, %16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7]), "s"(base)
      : "memory");
}

__device__ inline void st_cg_x4(void* p, u32x4 v) {
  // s_nop: a store of more than 64 bits reads its data registers over several cycles, and the compiler, which does not know that
  // this is one, may overwrite them in the very next instruction (seen in round 3: a 16-byte store written this way inside K4j's
  // tile loop stored the NEXT values; the slab loops below only got away with it because an LDS read sits in between)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" : : "v"(p), "v"(v) : "memory");
}
__device__ inline u32x4 ld_cg_x4(const void* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(kJThreads) void k_gc_joint_hist(
    const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits, int64_t n, int64_t nwords, unsigned int* __restrict__ slabs,
    unsigned int* __restrict__ tot /* kJTotWords, zero before the launch */, int per_group, unsigned int* __restrict__ counters,
    GcAccum* __restrict__ acc, double* __restrict__ table, uint8_t* __restrict__ d8, uint32_t* __restrict__ ghist,
    ValueHistAux* __restrict__ aux, ValueMedian* __restrict__ vm, const void* head_src, void* head_dst, unsigned int head_bytes, int dbg,
    unsigned int* __restrict__ esc_list /* gridDim.x lists of 1 + kJEscPerWg words */, unsigned int* __restrict__ rtab /* [202] */,
    JointInfo* __restrict__ info, unsigned int escape_limit, PhaseParams* __restrict__ pp, double cap_mult,
    NRuns nr /* K1b's work inside this launch (round 5): nr.list == NULL: the caller ran k_n_transitions */) {
  __shared__ __align__(16) unsigned int s_j[kJSlabWords];
  __shared__ unsigned int s_nesc;
  __shared__ WaveGc s_gc[kJWaves];
  __shared__ unsigned int s_flag, s_hi;
  __shared__ unsigned int s_bits[32];   // blocks of the pair counters that hold a count (kJBitWords words; the rest stay zero)
  for (int e = threadIdx.x; e < kJSlabWords; e += kJThreads) s_j[e] = 0;
  if (threadIdx.x < 32) s_bits[threadIdx.x] = 0u;
  if (threadIdx.x == 0) s_nesc = 0u;
  __syncthreads();
  // K1b's scan of the N mask, spread over this launch's workgroups (one launch less in the per-base phase: in a pool a launch costs
  // the phase 0.1-0.2 ms whatever it does); the list is complete when the last workgroup runs the tail
  if (nr.list) n_transitions_scan(nr.nbits, nwords, nr.list, nr.count, nr.cap);
  unsigned int* const my_list = esc_list + (size_t)blockIdx.x * (1 + kJEscPerWg);
  unsigned int* s_zero = s_j + kJOffZero;
  unsigned int* s_esc_cnt = s_j + kJOffEscCnt;
  unsigned long long* s_esc_sum = reinterpret_cast<unsigned long long*>(s_j + kJOffEscSum);
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  WaveGc& G = s_gc[wave];
  unsigned int nfast = 0, escapes = 0, negs = 0;   // per lane: values counted in the 16-bit fields; depths of 255 and more; negative depths
  // word of the pair counter of (level g, value v): the pair index is XOR-ed with the level's low bits, so that lanes whose
  // levels differ spread over the LDS banks (a row is 128 words: without it the bank would be a function of the value alone)
  const unsigned swz = (dbg & 4) ? 0u : 31u;
  auto cell = [&](uint32_t g, int v) -> unsigned int* { return &s_j[g * kJRowWords + (((unsigned)v >> 1) ^ (g & swz))]; };
  const bool no_atomics = (dbg & 2) != 0;
  const int64_t nsub = (n + kSubBases - 1) / kSubBases;
  const int64_t stride = (int64_t)gridDim.x * kJWaves;
  auto sat8 = [](int v) -> uint32_t { return v < 0 ? 0u : (v >= kByteEscape ? (uint32_t)kByteEscape : (uint32_t)v); };
  // one value that is not in 1 .. 254: depth 0 (32-bit counter per level), an escape, or a negative depth (refused by the host)
  auto odd_value = [&](int val, uint32_t g, int64_t pos) {
    if (val == 0) atomicAdd(&s_zero[g], 1u);
    else if (val > 0) {
      atomicAdd(&s_esc_cnt[g], 1u); atomicAdd(&s_esc_sum[g], (unsigned long long)val); ++escapes;
      const unsigned int k = atomicAdd(&s_nesc, 1u);             // the workgroup's first few escapes by position, for the last workgroup
      if (k < (unsigned int)kJEscPerWg) st_cg(&my_list[1 + k], (unsigned int)pos);
    }
    else negs = 1u;
  };

  // The streaming loop takes the INTERIOR sub-tiles (no window clamped at a chromosome end: all but the first and the last one
  // or two); its body has no branch around a global load or store -- every wave makes the same number of trips, a trip without
  // a sub-tile of its own repeats a harmless one with increments of zero and its byte copy sent to the array's padding -- so
  // that the compiler's wait counts are exact: with the edge sub-tiles' per-element loop and a conditional request inside, every
  // trip began by waiting for ALL memory operations in flight, the previous trip's store included (a memory round trip per trip
  // in the open).  The edge sub-tiles follow behind the loop.
  const int64_t first_in = 1;                                             // sub-tile 0 starts at base 0 < 101
  int64_t end_in = (n - 101) / kSubBases;                                 // interior: base + 1023 <= n - 102  <=>  sub < (n - 101) / 1024
  end_in = end_in < first_in ? first_in : end_in;
  const int64_t pad8 = ((n + kSubBases - 1) / kSubBases) * kSubBases;     // the byte copy's padding: n + 2048 bytes are allocated
  auto interior_trip = [&](const SubRegs& cur, int64_t sub, bool mine) {
    const int64_t base = sub * kSubBases;
    const int64_t first_bit = base - kGcLeft * 64;
    {
      const int64_t w = base / 64 - kGcLeft + lane;
      const uint64_t word = (lane < kSubLds && w >= 0 && w < nwords) ? cur.gw : 0;
      if (lane < kSubLds + 1) G.word[lane] = word;
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t i0 = base + 16 * (int64_t)lane;
    const uint32_t rel = (uint32_t)(i0 - 100 - first_bit);
    uint32_t cnt = wgc_window(G, rel);
    const uint32_t leave = wgc_field16(G, rel), enter = wgc_field16(G, rel + 201);
    const int v[16] = {cur.q[0].x, cur.q[0].y, cur.q[0].z, cur.q[0].w, cur.q[1].x, cur.q[1].y, cur.q[1].z, cur.q[1].w,
                       cur.q[2].x, cur.q[2].y, cur.q[2].z, cur.q[2].w, cur.q[3].x, cur.q[3].y, cur.q[3].z, cur.q[3].w};
    unsigned bad = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) bad |= (unsigned)((unsigned)(v[j] - 1) >= 254u);
    if (!bad) {   // the common case, straight-line: sixteen LDS atomics into [GC level][value pair]
      const unsigned int one = mine ? 1u : 0u;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (!no_atomics) atomicAdd(cell(cnt, v[j]), one << ((v[j] & 1) << 4));
        cnt = cnt - ((leave >> j) & 1u) + ((enter >> j) & 1u);
      }
      nfast += mine ? 16u : 0u;
    } else if (mine) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if ((unsigned)(v[j] - 1) < 254u) { atomicAdd(cell(cnt, v[j]), 1u << ((v[j] & 1) << 4)); ++nfast; }
        else odd_value(v[j], cnt, i0 + j);
        cnt = cnt - ((leave >> j) & 1u) + ((enter >> j) & 1u);
      }
    }
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = sat8(v[4 * k]) | (sat8(v[4 * k + 1]) << 8) | (sat8(v[4 * k + 2]) << 16) | (sat8(v[4 * k + 3]) << 24);
    *reinterpret_cast<uint4*>(d8 + (mine ? i0 : pad8 + 16 * (int64_t)lane)) = make_uint4(w[0], w[1], w[2], w[3]);
    __builtin_amdgcn_wave_barrier();   // the slot is rewritten by the next trip
  };
  auto edge_subtile = [&](int64_t sub) {   // the reference's clamped windows (App. A Q1), whole quads only, straight from memory
    const int64_t base = sub * kSubBases;
    const int64_t first_bit = base - kGcLeft * 64;
    {
      const int64_t w = base / 64 - kGcLeft + lane;
      if (lane < kSubLds + 1) G.word[lane] = (lane < kSubLds && w >= 0 && w < nwords) ? gcbits[w] : 0ull;
    }
    __builtin_amdgcn_wave_barrier();
    const int64_t i0 = base + 16 * (int64_t)lane;
    for (int j = 0; j < 16; ++j) {
      const int64_t i = i0 + j;
      if ((i & ~(int64_t)3) + 4 > n) break;
      int64_t lo = i - 100;
      if (lo < 0) lo = 0;
      if (lo > n - 202) lo = n - 202;
      const uint32_t g = wgc_window(G, (uint32_t)(lo - first_bit));
      const int val = depth[i];
      if ((unsigned)(val - 1) < 254u) { atomicAdd(cell(g, val), 1u << ((val & 1) << 4)); ++nfast; }
      else odd_value(val, g, i);
      d8[i] = (uint8_t)sat8(val);
    }
    __builtin_amdgcn_wave_barrier();
  };

  {
    const int64_t gw = (int64_t)blockIdx.x * kJWaves + wave;                 // this wave's number among all
    const int64_t nin = end_in - first_in;                                    // interior sub-tiles
    int64_t ntrips = (nin + stride - 1) / stride;
    ntrips += ntrips & 1;                                                     // two register sets used alternately: an even number of trips
    const int64_t safe = nin > 0 ? first_in : 0;                              // what a trip without a sub-tile of its own repeats (sub-tile 0 if there is no interior one: n >= 4040, so its loads are in range)
    auto sub_of = [&](int64_t t, bool& mine) { const int64_t sdx = first_in + gw + t * stride; mine = sdx < end_in; return mine ? sdx : safe; };
    SubRegs ra, rb;
    bool ma, mb;
    int64_t sa = sub_of(0, ma), sb = 0;
    sub_request(ra, depth, gcbits, nwords, sa * kSubBases, n, lane);
    for (int64_t t = 0; t < ntrips; t += 2) {
      sb = sub_of(t + 1, mb);
      sub_request(rb, depth, gcbits, nwords, sb * kSubBases, n, lane);
      interior_trip(ra, sa, ma && nin > 0);
      sa = sub_of(t + 2, ma);
      sub_request(ra, depth, gcbits, nwords, sa * kSubBases, n, lane);
      interior_trip(rb, sb, mb && nin > 0);
    }
    // the edge sub-tiles: dealt round robin to the waves from the far end of the grid (the near end's waves carry the remainder
    // of the interior ones)
    const int64_t nedge = first_in + (nsub - end_in);
    for (int64_t e = (stride - 1 - gw); e < nedge; e += stride) edge_subtile(e < first_in ? e : end_in + (e - first_in));
  }
  __syncthreads();
  // ---- did a 16-bit field wrap?  The fields must add up to what the lanes counted into them ----
  {
    unsigned int have = 0;
    for (int e = threadIdx.x; e < kJPacked; e += kJThreads) {   // whole waves (kJPacked is a multiple of 64): a wave's 64 words are two blocks
      const unsigned int w = s_j[e];
      have += (w & 0xffffu) + (w >> 16);
      const unsigned long long nz = __ballot(w != 0u);
      if (lane == 0 && nz != 0ull) {
        const int b = e >> 5;
        if (nz & 0xffffffffull) atomicOr(&s_bits[b >> 5], 1u << (b & 31));
        if (nz >> 32) atomicOr(&s_bits[(b + 1) >> 5], 1u << ((b + 1) & 31));
      }
    }
    unsigned int diff = nfast - have;   // modulo 2^32: zero over the workgroup iff nothing wrapped (a workgroup sees < 2^31 bases)
    for (int d = 32; d >= 1; d >>= 1) { diff += __shfl_xor(diff, d); escapes += __shfl_xor(escapes, d); negs |= __shfl_xor(negs, d); }
    if (threadIdx.x == 0) s_flag = 0u;
    __syncthreads();
    if (lane == 0) { atomicAdd(&s_flag, diff); if (escapes) atomicAdd(&s_j[kJOffFlags + 1], escapes); if (negs) atomicOr(&s_j[kJOffFlags + 0], 1u); }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (s_flag != 0u) s_j[kJOffFlags + 2] = 1u;
      st_cg(&my_list[0], s_nesc);                                   // how many escapes the workgroup saw (listed: the first kJEscPerWg)
      if (s_nesc > (unsigned int)kJEscPerWg) s_j[kJOffFlags + 3] = 1u;   // workgroups whose list ran over
    }
    __syncthreads();
  }
  if (dbg & 1) { if (blockIdx.x == 0) export_words(head_dst, head_src, head_bytes); return; }   // DEBUG ablation: no slab, no fold, no tail
  // ---- slab out (16-byte write-through stores), then the two-level hand-over of device_util.h with a fold of its own: the
  // last workgroup of a group unpacks the group's slabs into 32-bit sums and adds them to the totals ----
  {
    unsigned int* slab = slabs + (size_t)blockIdx.x * kJSlabStride;
    for (int q = threadIdx.x; q < kJSlabWords / 4; q += kJThreads) {
      if (4 * q < kJPacked && !((s_bits[q >> 8] >> ((q >> 3) & 31)) & 1u)) continue;   // block q / 8 holds no count: not written
      u32x4 v; v.x = s_j[4 * q]; v.y = s_j[4 * q + 1]; v.z = s_j[4 * q + 2]; v.w = s_j[4 * q + 3];
      st_cg_x4(slab + 4 * q, v);
    }
    if (threadIdx.x < 8) {   // the bitmap and the zero words behind it
      u32x4 v; v.x = s_bits[4 * threadIdx.x]; v.y = s_bits[4 * threadIdx.x + 1]; v.z = s_bits[4 * threadIdx.x + 2]; v.w = s_bits[4 * threadIdx.x + 3];
      st_cg_x4(slab + kJSlabWords + 4 * threadIdx.x, v);
    }
  }
  const int nblocks = (int)gridDim.x;
  const int grp = (int)blockIdx.x / per_group;
  const int ngroups = (nblocks + per_group - 1) / per_group;
  const int members = (grp + 1) * per_group <= nblocks ? per_group : nblocks - grp * per_group;
  drain();
  __syncthreads();   // the slab's stores have completed
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(&counters[1 + grp], 1u);
    const bool last = t == (unsigned int)members - 1u;
    if (last) atomicExch(&counters[1 + grp], 0u);
    s_flag = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_flag) return;
  {
    const unsigned int* src = slabs + (size_t)grp * per_group * kJSlabStride;
    const int last = __builtin_amdgcn_readfirstlane(members - 1);
    // the members' block bitmaps (the slab image in LDS is not needed any more), their union as a list of blocks
    unsigned int* m_bits = s_j;                 // [members][32]
    unsigned int* b_list = s_j + 32 * 32;       // up to kJBlocks entries
    for (int e = threadIdx.x; e < members * 32; e += kJThreads) m_bits[e] = ld_cg(src + (size_t)(e >> 5) * kJSlabStride + kJSlabWords + (e & 31));
    if (threadIdx.x == 0) s_hi = 0u;
    __syncthreads();
    for (int b = threadIdx.x; b < kJBlocks; b += kJThreads) {
      unsigned int any = 0;
      for (int k = 0; k <= last; ++k) any |= m_bits[k * 32 + (b >> 5)];
      if ((any >> (b & 31)) & 1u) b_list[atomicAdd(&s_hi, 1u)] = (unsigned int)b;
    }
    __syncthreads();
    const int nlist = (int)s_hi;
    const unsigned long long src_base = uniform_address(src);
    // ---- the pair counters: the quads of the listed blocks; a member that did not write a block is read at its zero quad ----
    for (int idx = threadIdx.x; idx < nlist * 8; idx += kJThreads) {
      const int b = (int)b_list[idx >> 3];
      const int q = b * 8 + (idx & 7);
      unsigned int lo4[4] = {0, 0, 0, 0}, hi4[4] = {0, 0, 0, 0};
      for (int k0 = 0; k0 <= last; k0 += 8) {
        unsigned int off[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + j < last ? k0 + j : last;
          const bool has = (m_bits[k * 32 + (b >> 5)] >> (b & 31)) & 1u;
          off[j] = ((unsigned int)k * (unsigned int)kJSlabStride + (has ? (unsigned int)q * 4u : (unsigned int)kJZeroQuad)) * 4u;
        }
        u32x4 v[8];
        ld_cg_x4_batch8(v, src_base, off);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + j <= last) {
          const unsigned int w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) { lo4[c] += w[c] & 0xffffu; hi4[c] += w[c] >> 16; }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ws = 4 * q + c;                                            // word g * 128 + (pair ^ (g & 31)) of the slab
        const int wi = (ws & ~127) | ((ws & 127) ^ (int)((ws >> 7) & swz));     // <-> counters 2 * wi, 2 * wi + 1 of the 32-bit table
        if (lo4[c]) atomicAdd(&tot[2 * wi], lo4[c]);
        if (hi4[c]) atomicAdd(&tot[2 * wi + 1], hi4[c]);
      }
    }
    // ---- behind them: the zero / escape counters per level, the flags, the escapes' 64-bit sums (always written) ----
    for (int q = kJPacked / 4 + threadIdx.x; q < kJSlabWords / 4; q += kJThreads) {
      unsigned int lo4[4] = {0, 0, 0, 0}, hi4[4] = {0, 0, 0, 0};
      for (int k0 = 0; k0 <= last; k0 += 8) {
        unsigned long long b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = uniform_address(src + (size_t)(k0 + j < last ? k0 + j : last) * kJSlabStride);
        u32x4 v[8];
        ld_cg_x8(v, (unsigned int)q * 16u, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k0 + j <= last) {
          const unsigned int w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) { const unsigned int o = lo4[c]; lo4[c] += w[c]; hi4[c] += lo4[c] < o; }   // plain words; hi4 = carries (the 64-bit sums' low halves)
        }
      }
      if (4 * q < kJOffEscSum) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (lo4[c]) atomicAdd(&tot[kJTotH + (4 * q + c - kJPacked)], lo4[c]);
      } else {   // the escapes' 64-bit sums: two per vector; each slab's value is below 2^63, eight of them are added as (low, high) halves
        unsigned long long* t64 = reinterpret_cast<unsigned long long*>(tot + kJTotH + (4 * q - kJPacked));
        const unsigned long long a = (unsigned long long)lo4[0] + ((unsigned long long)hi4[0] << 32) + ((unsigned long long)lo4[1] << 32);
        const unsigned long long c2 = (unsigned long long)lo4[2] + ((unsigned long long)hi4[2] << 32) + ((unsigned long long)lo4[3] << 32);
        if (a) atomicAdd(&t64[0], a);
        if (c2) atomicAdd(&t64[1], c2);
      }
    }
  }
  drain();
  __syncthreads();   // this group's atomics have completed
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(&counters[0], 1u);
    const bool last = t == (unsigned int)ngroups - 1u;
    if (last) atomicExch(&counters[0], 0u);
    s_flag = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_flag) return;

  if (dbg & 8) { export_words(head_dst, head_src, head_bytes); return; }   // DEBUG ablation: no tail
  // ================= the launch's last workgroup: GC table, rescaled-value histogram, cap median, header =================
  if (nr.list) {   // the removed regions K4s compacts with (K1b's tail): before anything below can return
    __shared__ unsigned long long s_re[kRegSortMax], s_rs[kRegSortMax];
    n_regions_build(nr.list, nr.count, n, nr.dx, pp, nr.cbreak, nr.cum, s_re, s_rs);
    __syncthreads();
  }
  // scratch in the (now free) slab image
  unsigned long long* r_sum = reinterpret_cast<unsigned long long*>(s_j);            // [202]
  unsigned long long* r_cnt = r_sum + kGcLevels;                                     // [202]
  double* r_tab = reinterpret_cast<double*>(r_cnt + kGcLevels);                      // [203]
  unsigned long long* r_misc = reinterpret_cast<unsigned long long*>(r_tab + kGcLevels + 2);   // [4]: possum, poscnt
  unsigned int* r_hist = reinterpret_cast<unsigned int*>(r_misc + 4);                // [kJRh]
  for (int e = threadIdx.x; e < kJRh; e += kJThreads) r_hist[e] = 0;
  for (int e = threadIdx.x; e < 2 * kGcLevels; e += kJThreads) r_sum[e] = 0;   // r_sum and r_cnt
  if (threadIdx.x == 0) { s_hi = 0u; r_misc[0] = 0; r_misc[1] = 0; }
  __syncthreads();
  static_assert(kJThreads >= kGcLevels, "thread g holds level g's counters");
  // ---- H into registers, once: row g (256 cells) is one 16-byte quad per lane, every wave takes rows wave, wave + 12, ...
  // (seventeen loads per lane, eight in flight).  Row sums (count, sum of depths per level) and the largest depth byte present
  // are wave reductions -- 64-bit LDS atomics from every quad were 20 of the tail's 55 us -- and the rows stay in registers for
  // the pass behind the table, which needs them again. ----
  __shared__ unsigned int s_vmax, s_gmin, s_gmax;
  if (threadIdx.x == 0) { s_vmax = 0u; s_gmin = 0xffffffffu; s_gmax = 0u; }
  __syncthreads();
  constexpr int kRowsPerWave = (kGcLevels + kJWaves - 1) / kJWaves;   // 17
  u32x4 rowq[kRowsPerWave];
  const __amdgpu_buffer_rsrc_t tot_rs = coherent_buffer(tot);
#pragma unroll
  for (int k = 0; k < kRowsPerWave; ++k) {
    const int g = wave + k * kJWaves;
    rowq[k] = ld_cg_buf_x4(tot_rs, (unsigned int)(4 * lane) * 4u, (unsigned int)((g < kGcLevels ? g : 0) * 256) * 4u);
  }
  // the per-level counters behind H and the flags ride in the same burst (they were three and one more round trips further down)
  unsigned int pz = 0, pe = 0;
  unsigned long long pes = 0;
  u32x4 pflags = {0u, 0u, 0u, 0u};
  if (threadIdx.x < kGcLevels) {
    const unsigned int g4 = (unsigned int)threadIdx.x * 4u;
    pz = __builtin_amdgcn_raw_buffer_load_b32(tot_rs, (int)((unsigned int)(kJTotH + (kJOffZero - kJPacked)) * 4u + g4), 0, 16);
    pe = __builtin_amdgcn_raw_buffer_load_b32(tot_rs, (int)((unsigned int)(kJTotH + (kJOffEscCnt - kJPacked)) * 4u + g4), 0, 16);
    const auto e2 = __builtin_amdgcn_raw_buffer_load_b64(tot_rs, (int)((unsigned int)(kJTotH + (kJOffEscSum - kJPacked)) * 4u + 2u * g4), 0, 16);
    pes = (unsigned long long)e2[0] | ((unsigned long long)e2[1] << 32);
  }
  if (threadIdx.x == 0) pflags = ld_cg_buf_x4(tot_rs, (unsigned int)(kJTotH + (kJOffFlags - kJPacked)) * 4u, 0u);
  {
    unsigned int wave_top = 0;
#pragma unroll
    for (int k = 0; k < kRowsPerWave; ++k) {
      const int g = wave + k * kJWaves;
      if (g >= kGcLevels) continue;                       // wave-uniform
      const unsigned int c[4] = {rowq[k].x, rowq[k].y, rowq[k].z, rowq[k].w};
      const int v0 = 4 * lane;
      unsigned int cs = c[0] + c[1] + c[2] + c[3];          // a level holds fewer than 2^31 bases
      unsigned long long vs = 0;
      unsigned int top = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { vs += (unsigned long long)c[j] * (unsigned long long)(v0 + j); top = c[j] ? (unsigned int)(v0 + j) : top; }
      // wave totals in lane 63 (DPP scans: no trips through the LDS crossbar in this one-workgroup tail)
      cs = (unsigned int)wave_lane63(wave_incl_scan((int)cs));
      vs = (unsigned long long)wave_lane63(wave_incl_scan((long long)vs));
      top = (unsigned int)wave_lane63((int)wave_incl_max(top));
      if (lane == 0) { r_cnt[g] = cs; r_sum[g] = vs; }
      wave_top = top > wave_top ? top : wave_top;
    }
    if (lane == 0 && wave_top) atomicMax(&s_vmax, wave_top);
  }
  __syncthreads();
  for (int g = threadIdx.x; g < kGcLevels; g += kJThreads) {
    const unsigned long long z = pz, ec = pe, es = pes;   // g == threadIdx.x: kJThreads >= kGcLevels
    const unsigned long long vs = r_sum[g], cs = r_cnt[g];
    r_sum[g] = vs + es;
    r_cnt[g] = cs + z + ec;
    atomicAdd(&r_misc[0], vs + es);          // depth > 0: every counted value but the zeros
    atomicAdd(&r_misc[1], cs + ec);
    if (cs + z + ec) { atomicMin(&s_gmin, (unsigned int)g); atomicMax(&s_gmax, (unsigned int)g); }
  }
  __syncthreads();
  // ---- the ragged last n % 4 bases: the stale-window zone i >= n-101, count of [n-202, n-2] (App. A Q1); table only (the value
  // histogram gets them from gc_tail_fixup, as raw depths) ----
  __shared__ unsigned int s_esc_total, s_list_over, s_bad_flags;
  if (threadIdx.x == 0) {
    const u32x4 f4 = pflags;   // negative depth seen | escapes | a workgroup's counters wrapped | a list ran over
    unsigned int fl_neg = f4.x;
    unsigned long long esc = f4.y;
    if ((n & 3) != 0) {
      const int g = gc_window_count(gcbits, n - 202);
      for (int64_t i = n & ~(int64_t)3; i < n; ++i) {
        const int v = depth[i];
        r_sum[g] += (unsigned long long)(long long)v;
        r_cnt[g] += 1ull;
        if (v > 0) { r_misc[0] += (unsigned long long)v; r_misc[1] += 1ull; }
        if (v < 0) fl_neg |= 1u;
        d8[i] = (uint8_t)sat8(v);
        esc += v >= kByteEscape;
      }
    }
    acc->possum = r_misc[0]; acc->poscnt = r_misc[1];
    s_bad_flags = (fl_neg ? 1u : 0u) | (f4.z ? 4u : 0u);
    acc->negatives = s_bad_flags;   // bit 2: a workgroup's 16-bit counters wrapped, nothing below is valid
    const unsigned int esc32 = esc > 0xffffffffull ? 0xffffffffu : (unsigned int)esc;
    acc->escapes = esc32;
    s_esc_total = esc32;
    s_list_over = f4.w;
  }
  __syncthreads();
  const unsigned long long ps = r_misc[0], pc = r_misc[1];
  double rdmean = (double)ps;
  if (pc > 0) rdmean /= (double)pc;
  for (int g = threadIdx.x; g < kGcLevels; g += kJThreads) {
    const unsigned long long sg = r_sum[g], cg = r_cnt[g];
    acc->sum[g] = sg; acc->cnt[g] = cg;
    double t = cg > 0 ? (double)sg / (double)cg : rdmean;
    if (t < 1) t = rdmean;
    table[g] = t;
    r_tab[g] = t;
  }
  if (threadIdx.x == 0) { table[kGcLevels] = rdmean; r_tab[kGcLevels] = rdmean; }
  __syncthreads();
  // what K4j needs to know about the table it rescales with: the levels and depth bytes that occur, whether the histogram is complete
  const bool deep = s_esc_total > escape_limit;                    // the host takes the int32 kernels: nothing more to do here
  const bool esc_pending = s_esc_total != 0u && (s_list_over != 0u || deep);
  if (threadIdx.x == 0) {
    info->gmin = s_gmin > s_gmax ? 0 : (int)s_gmin; info->gmax = s_gmin > s_gmax ? 0 : (int)s_gmax;
    info->vmax = (int)s_vmax; info->esc_pending = esc_pending ? 1 : 0;
  }
  if (threadIdx.x == 0) { pp->capval = -1; pp->redo = 0; }         // until the median is known (and when it will not be: K4j then declines)
  if (deep || (dbg & 16)) { sync_drained(); export_words(head_dst, head_src, head_bytes); return; }   // (dbg 16: ablation, table only)
  // ---- K4j's rescale without floating point: per level a fixed-point ratio R (22 fraction bits: 2^-23 * 254 is the distance from
  // a rounding boundary at which it can go wrong) with (v * R + 2^21) >> 22 == the reference's (int)(v * rdmean / table[g] + 0.5)
  // for EVERY depth byte v = 0 .. 254 (compared saturated at kByteSat: K4j caps below that) -- checked here, cell by cell,
  // against the reference's expression; a level where some byte disagrees, or whose ratio is 4 and more (R must stay below
  // 2^24: the product is a 24 x 8 bit multiply in 32 bits), carries bit 31 and sends its bases through the exact expression.
  unsigned int* r_bad = reinterpret_cast<unsigned int*>(r_hist + kJRh);   // [202]
  for (int g = threadIdx.x; g < kGcLevels; g += kJThreads) r_bad[g] = 0u;
  __syncthreads();
  // ---- one pass over the rows in registers: the reference's expression once per (level, depth byte) cell, compared with the
  // fixed-point form (every depth byte of every level that occurs) and, where the cell is not empty, counted into the histogram
  // of the rescaled depth ----
  unsigned int lane_hi = 0;
  auto count_value = [&](int r, unsigned int c) {
    if (r >= 0 && r < kJRh) atomicAdd(&r_hist[r], c);
    else if (r >= 0 && r < kHistValues) { atomicAdd(&ghist[r], c); lane_hi = (unsigned)r > lane_hi ? (unsigned)r : lane_hi; }
    else if (r < 0) atomicOr(&aux->negatives, 1u);
    else { atomicAdd(&aux->big, (unsigned long long)c); lane_hi = 0xffffffffu; }
  };
#pragma unroll
  for (int k = 0; k < kRowsPerWave; ++k) {
    const int g = wave + k * kJWaves;
    if (g >= kGcLevels || r_cnt[g] == 0) continue;        // wave-uniform
    const unsigned int c[4] = {rowq[k].x, rowq[k].y, rowq[k].z, rowq[k].w};
    const double tg = r_tab[g];
    const double ratio = rdmean / tg;
    bool bad = !(ratio < kFixMaxRatio);
    const unsigned int R = bad ? 0u : (unsigned int)(ratio * (double)(1u << kFixShift) + 0.5);
    // The reference's (int)(v * rdmean / tg + 0.5) without a division per cell: v * rdmean times the row's reciprocal is within a few
    // ulp of the rounded quotient (< 1e-10 here: the quotient is below 2^17), so wherever that sum lies further than 1e-6 from an
    // integer both truncate alike; a cell that does not (one in a million) takes the division.  (The division was a third of the
    // tail: 68 of them per lane, 35 double-precision instructions each.)
    const double inv_tg = 1.0 / tg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = 4 * lane + j;
      if (v >= kByteEscape) continue;
      const double t1 = (double)v * rdmean;
      const double sa = t1 * inv_tg + 0.5;
      const double fr = __builtin_amdgcn_fract(sa);
      int r = (int)sa;
      if (fr < 1e-6 || fr > 1.0 - 1e-6) r = (int)(t1 / tg + 0.5);   // gccontent.cpp:89, truncation
      const int rs = r > kByteSat ? kByteSat : r;
      unsigned int f = ((unsigned int)v * R + (1u << (kFixShift - 1))) >> kFixShift;
      f = f > (unsigned int)kByteSat ? (unsigned int)kByteSat : f;
      bad = bad || f != (unsigned int)rs;
      if (c[j]) count_value(r, c[j]);
    }
    if (__ballot(bad) != 0ull && lane == 0) r_bad[g] = 1u;   // the row's only writer
  }
  __syncthreads();
  for (int g = threadIdx.x; g < kGcLevels; g += kJThreads) {
    const double ratio = rdmean / r_tab[g];
    const bool wide = !(ratio < kFixMaxRatio);
    // bit 31: no verified ratio (take the exact expression); bit 30: the level does not occur at all (K4s: nothing to look out for)
    rtab[g] = (wide ? 0u : (unsigned int)(ratio * (double)(1u << kFixShift) + 0.5)) | ((r_bad[g] || wide || r_cnt[g] == 0) ? 0x80000000u : 0u) | (r_cnt[g] == 0 ? 0x40000000u : 0u);
  }
  {
    if (threadIdx.x < kGcLevels && pz) atomicAdd(&r_hist[0], pz);   // (int)(0 * ratio + 0.5) = 0
    // the escapes the workgroups listed by position: the reference's expression on the int32 depth, window by the clamped rule
    if (s_esc_total != 0u && !esc_pending) {
      const int nlists = (int)gridDim.x;
      for (int sidx = threadIdx.x; sidx < nlists * kJEscPerWg; sidx += kJThreads) {
        const int w = sidx / kJEscPerWg, k = sidx - w * kJEscPerWg;
        const unsigned int* L = esc_list + (size_t)w * (1 + kJEscPerWg);
        if ((unsigned int)k >= ld_cg(L)) continue;
        const int64_t i = (int64_t)ld_cg(L + 1 + k);
        int64_t lo = i - 100;
        if (lo < 0) lo = 0;
        if (lo > n - 202) lo = n - 202;
        count_value((int)((double)depth[i] * rdmean / r_tab[gc_window_count(gcbits, lo)] + 0.5), 1u);
      }
    }
    for (int d = 32; d >= 1; d >>= 1) { const unsigned int o = (unsigned int)__shfl_xor((int)lane_hi, d); lane_hi = o > lane_hi ? o : lane_hi; }
    if (lane == 0 && lane_hi) atomicMax(&s_hi, lane_hi);
  }
  __syncthreads();
  // ---- histogram side of the tail quirks of the 20-slice write-back (App. A Q2/Q3) and the ragged bases, as K3' does it: on the
  // LDS counters where the values lie inside them, else on the global histogram (marking s_hi, which sends the rest of the
  // tail down the global road) ----
  if (threadIdx.x < 64) {
    auto t_add = [&](int to) {
      if (to >= 0 && to < kJRh) atomicAdd(&r_hist[to], 1u);
      else if (to >= 0 && to < kHistValues) { atomicAdd(&ghist[to], 1u); atomicMax(&s_hi, (unsigned int)to); }
      else if (to >= kHistValues) { atomicAdd(&aux->big, 1ull); atomicMax(&s_hi, 0xffffffffu); }
      else { atomicOr(&aux->negatives, 1u); atomicMax(&s_hi, 0xffffffffu); }
    };
    auto t_sub = [&](int from) {
      if (from >= 0 && from < kJRh) atomicSub(&r_hist[from], 1u);
      else if (from >= 0 && from < kHistValues) { atomicSub(&ghist[from], 1u); atomicMax(&s_hi, (unsigned int)from); }
      else if (from >= kHistValues) { atomicAdd(&aux->big, (unsigned long long)-1ll); atomicMax(&s_hi, 0xffffffffu); }
    };
    gc_tail_fixup_with(depth, gcbits, n, r_tab, 1, nullptr, t_add, t_sub);
  }
  __syncthreads();
  // ---- the walk to the median apply_cap needs (loaddata.cpp:233; partition_stat_tp's walk, wufunctions.cpp:398-420, dy = 1).
  // Every rescaled value below kJRh (all of them, on any ordinary chromosome): the histogram never leaves LDS -- counts, the
  // bucket where the running count reaches n / 2, smallest and largest value, the cap for a queued K4j, all from registers
  // and LDS; what used to be eight global round trips (flush, drain, fix-up, drain, reload, walk, reload, reload).  Nobody reads
  // the global histogram of a chromosome without pending escapes, so it is not written either. ----
  const bool lds_walk = !esc_pending && s_hi == 0u;
  __shared__ unsigned long long s_wsum[kJWaves];
  __shared__ int s_wlo[kJWaves], s_whi[kJWaves], s_wmed;
  if (lds_walk) {
    constexpr int kPer = (kJRh + kJThreads - 1) / kJThreads;   // 11 consecutive counters per thread
    const int v0 = threadIdx.x * kPer;
    unsigned int c[kPer];
    unsigned long long local = 0;
    int lo = 0x7fffffff, hi = -1;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      c[i] = v0 + i < kJRh ? r_hist[v0 + i] : 0u;
      local += c[i];
      if (c[i]) { lo = v0 + i < lo ? v0 + i : lo; hi = v0 + i; }
    }
    const unsigned long long incl = (unsigned long long)wave_incl_scan((long long)local);
    int wlo = lo, whi = hi;
    for (int d = 32; d >= 1; d >>= 1) { const int a = __shfl_xor(wlo, d), b2 = __shfl_xor(whi, d); wlo = a < wlo ? a : wlo; whi = b2 > whi ? b2 : whi; }
    if (lane == 63) s_wsum[wave] = incl;
    if (lane == 0) { s_wlo[wave] = wlo; s_whi[wave] = whi; }
    if (threadIdx.x == 0) s_wmed = -1;
    __syncthreads();
    unsigned long long before = 0, all = 0;
    int glo = 0x7fffffff, ghi = -1;
    for (int w = 0; w < kJWaves; ++w) { if (w < wave) before += s_wsum[w]; all += s_wsum[w]; glo = s_wlo[w] < glo ? s_wlo[w] : glo; ghi = s_whi[w] > ghi ? s_whi[w] : ghi; }
    const unsigned long long r2 = (unsigned long long)n / 2;
    unsigned long long seen = before + incl - local;
    if (local != 0 && seen < r2 && seen + local >= r2) {   // the walk's bucket lies in this thread's stretch
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const unsigned long long upto = seen + c[i];
        if (seen < r2 && upto >= r2) s_wmed = v0 + i;
        seen = upto;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int med = s_wmed;
      st_cg(&vm->inrange, all);
      st_cg(reinterpret_cast<unsigned int*>(&vm->lo), (unsigned int)glo); st_cg(reinterpret_cast<unsigned int*>(&vm->hi), (unsigned int)ghi);
      st_cg(reinterpret_cast<unsigned int*>(&vm->med), (unsigned int)med); st_cg(reinterpret_cast<unsigned int*>(&vm->pad), 0u);
      // the cap as apply_cap takes it (loaddata.cpp:233-238: median of the uncompacted array, RD = median * cap truncated), for a
      // K4j queued right behind this launch; the host derives the same number from the header and checks everything else
      if (cap_mult > 1.0) {
        double qm = (double)glo;
        if (glo <= ghi && (double)ghi - (double)glo >= 1.0 && med >= 0) qm = (double)med;
        const bool ok = all == (unsigned long long)n && (unsigned long long)n / 2 <= all && s_bad_flags == 0u;   // (nothing went to aux->big on this road)
        pp->capval = ok ? (int32_t)(qm * cap_mult) : -1;
      }
    }
    sync_drained();
    export_words(head_dst, head_src, head_bytes);
    return;
  }
  // ---- the global road: values beyond the LDS counters, or escapes still to come (k_escape_hist adds them and walks again) ----
  for (int e = threadIdx.x; e < kJRh; e += kJThreads) { const unsigned int c = r_hist[e]; if (c) { atomicAdd(&ghist[e], c); atomicMax(&s_hi, (unsigned int)e); } }
  sync_drained();
  if (threadIdx.x == 0 && s_hi >= (unsigned int)kValLds) atomicMax(&aux->vmax, s_hi >= (unsigned int)kHistValues ? (unsigned int)kHistValues : s_hi);
  sync_drained();
  {
    const unsigned int hi = ld_cg(&aux->vmax);
    int range = hi >= (unsigned int)kHistValues ? kHistValues : (int)hi + 1;
    const int top = (int)(s_hi >= (unsigned int)kHistValues ? (unsigned int)kHistValues - 1u : s_hi) + 1;
    range = range < top ? top : range;
    range = range < kValLds ? kValLds : range;
    value_median_block<kJThreads>(ghist, (unsigned long long)n, vm, range > kHistValues ? kHistValues : range);
  }
  sync_drained();
  if (threadIdx.x == 0 && !esc_pending && cap_mult > 1.0) {
    const int lo = (int)ld_cg(reinterpret_cast<const unsigned int*>(&vm->lo)), hi = (int)ld_cg(reinterpret_cast<const unsigned int*>(&vm->hi));
    const int med = (int)ld_cg(reinterpret_cast<const unsigned int*>(&vm->med));
    double qm = (double)lo;
    if (lo <= hi && (double)hi - (double)lo >= 1.0 && med >= 0) qm = (double)med;
    const unsigned long long inr = ld_cg(&vm->inrange);
    const bool ok = inr + aux->big == (unsigned long long)n && (unsigned long long)n / 2 <= inr && s_bad_flags == 0u;
    pp->capval = ok ? (int32_t)(qm * cap_mult) : -1;
  }
  sync_drained();
  export_words(head_dst, head_src, head_bytes);
}

// The escapes' share of the value histogram (depths of 255 and more: the byte copy says kByteEscape): K2j's table is known,
// so each of them is rescaled with the reference's expression and counted.  Launched only when K2j's header reports escapes
// (and fewer than the byte path's limit); its last workgroup walks the histogram to the cap median and hands the header over
// once more.
__global__ __launch_bounds__(kThreads) void k_escape_hist(const uint8_t* __restrict__ d8, const int32_t* __restrict__ depth,
                                                          const uint64_t* __restrict__ gcbits, int64_t n, const double* __restrict__ table,
                                                          uint32_t* __restrict__ ghist, ValueHistAux* __restrict__ aux,
                                                          unsigned int* __restrict__ counter, ValueMedian* __restrict__ vm,
                                                          const void* head_src, void* head_dst, unsigned int head_bytes) {
  const double rdmean = table[kGcLevels];
  const int64_t whole = n & ~(int64_t)3;   // the ragged bases belong to the tail fixup (raw depths), as everywhere
  int lane_hi = 0;
  for (int64_t c = (int64_t)blockIdx.x * kThreads + threadIdx.x; c * 16 < whole; c += (int64_t)gridDim.x * kThreads) {
    const uint4 b = *reinterpret_cast<const uint4*>(d8 + 16 * c);   // the copy is padded past n
    if (!(has_escape(b.x) || has_escape(b.y) || has_escape(b.z) || has_escape(b.w))) continue;
    const uint32_t w4[4] = {b.x, b.y, b.z, b.w};
    for (int j = 0; j < 16; ++j) {
      const int64_t i = 16 * c + j;
      if (i >= whole || ((w4[j >> 2] >> (8 * (j & 3))) & 0xffu) != (uint32_t)kByteEscape) continue;
      int64_t lo = i - 100;
      if (lo < 0) lo = 0;
      if (lo > n - 202) lo = n - 202;
      const int g = gc_window_count(gcbits, lo);
      const int r = (int)((double)depth[i] * rdmean / table[g] + 0.5);   // gccontent.cpp:89, truncation
      if (r >= 0 && r < kHistValues) atomicAdd(&ghist[r], 1u);
      else if (r < 0) atomicOr(&aux->negatives, 1u);
      else atomicAdd(&aux->big, 1ull);
      lane_hi = r > lane_hi ? r : lane_hi;
    }
  }
  for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(lane_hi, d); lane_hi = o > lane_hi ? o : lane_hi; }
  if (lane_id() == 0 && lane_hi >= kValLds) atomicMax(&aux->vmax, (unsigned int)(lane_hi >= kHistValues ? kHistValues : lane_hi));
  if (!last_block_done(counter)) return;
  {
    const unsigned int hi = ld_cg(&aux->vmax);
    int range = hi >= (unsigned int)kHistValues ? kHistValues : (int)hi + 1;
    range = range < kValLds ? kValLds : range;
    value_median_block<kThreads>(ghist, (unsigned long long)n, vm, range);
  }
  sync_drained();
  export_words(head_dst, head_src, head_bytes);
}

// ------------------------------------------------------------------------------------------
// K3  gc_rescale (+ value histogram for the cap median)
template <bool ADJUST>
__global__ __launch_bounds__(kThreads) void k_gc_rescale(const int32_t* __restrict__ depth,
                                                         const uint64_t* __restrict__ gcbits, int64_t n,
                                                         int64_t nwords, const double* __restrict__ table /* [kGcLevels] + rdmean */,
                                                         int32_t* __restrict__ out,
                                                         uint32_t* __restrict__ ghist, ValueHistAux* __restrict__ aux,
                                                         unsigned int* __restrict__ hist_slabs, unsigned int* __restrict__ gsum,
                                                         int per_group, unsigned int* __restrict__ counters,
                                                         ValueMedian* __restrict__ vm, const void* head_src, void* head_dst,
                                                         unsigned int head_bytes, int materialize /* 1: only out[] is produced */,
                                                         uint8_t* __restrict__ out8 /* optional: the values as bytes, saturated at kByteSat (-NOGC: what K4' compacts) */,
                                                         PhaseParams* __restrict__ pp /* optional: the cap for a K4 queued behind this launch */, double cap_mult) {
  __shared__ GcTile gt;
  __shared__ __align__(16) unsigned char s_g[ADJUST ? kTileBases : 16];
  if (pp && blockIdx.x == 0 && threadIdx.x == 0) {   // until the median is known (and when it will not be: the queued K4 then declines)
    st_cg(reinterpret_cast<unsigned int*>(&pp->capval), 0xffffffffu); st_cg(reinterpret_cast<unsigned int*>(&pp->redo), 0u);
  }
  __shared__ double s_table[kGcLevels];
  __shared__ unsigned int s_hist[kValLds * 32];
  for (int e = threadIdx.x; e < kValLds * 32; e += kThreads) s_hist[e] = 0;
  if (ADJUST) for (int e = threadIdx.x; e < kGcLevels; e += kThreads) s_table[e] = table[e];
  const double rdmean = ADJUST ? table[kGcLevels] : 0.0;
  const bool deep = ADJUST && rdmean >= 160.0;                // deep coverage: the LDS histogram follows the distribution
  const int width = deep ? kDeepWidth : kValLds, phsh = deep ? kDeepPhaseShift : 5;
  const int vb = deep ? hist_window_base_dev(rdmean, width) : 0;
  const int phase = threadIdx.x & ((1 << phsh) - 1);
  int lane_hi = 0;   // largest value this lane sent past the LDS range
  const int64_t ntiles = (n + kTileBases - 1) / kTileBases;
  auto trip = [&](const TileRegs& cur, const GcRegs& gcur, TileRegs& nxt, GcRegs& gnxt, int64_t tile) {
    const int64_t base = tile * kTileBases;
    const int64_t first_bit = base - kGcLeft * 64;
    if (ADJUST) {
      __syncthreads();
      gc_tile_commit(gt, gcur, nwords, base / 64);
      __syncthreads();
      tile_window_counts(gt, s_g, base, n, first_bit);
    }
    if (tile + gridDim.x < ntiles) {
      tile_request(nxt, depth, (tile + gridDim.x) * kTileBases, n);
      if (ADJUST) gc_tile_request(gnxt, gcbits, nwords, (tile + gridDim.x) * kGcWords);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kTileBases / (4 * kThreads); ++k) {
      const int li = 4 * (k * kThreads + threadIdx.x);
      const int64_t q = base + li;
      if (q + 4 > n) continue;
      const uint32_t g4 = ADJUST ? *reinterpret_cast<const uint32_t*>(s_g + li) : 0u;
      // named components only: a runtime-indexed array would live in scratch memory
      auto one = [&](int val, uint32_t g) -> int {
        if (ADJUST) val = (int)((double)val * rdmean / s_table[g] + 0.5);   // gccontent.cpp:89, truncation
        return val;
      };
      const int v0 = one(cur.q[k].x, g4 & 0xffu);
      const int v1 = one(cur.q[k].y, (g4 >> 8) & 0xffu);
      const int v2 = one(cur.q[k].z, (g4 >> 16) & 0xffu);
      const int v3 = one(cur.q[k].w, g4 >> 24);
      if (materialize) {
      } else if (((unsigned)(v0 - vb) | (unsigned)(v1 - vb) | (unsigned)(v2 - vb) | (unsigned)(v3 - vb)) < (unsigned)width) {   // the common case, branch-free
        atomicAdd(&s_hist[((v0 - vb) << phsh) + phase], 1u); atomicAdd(&s_hist[((v1 - vb) << phsh) + phase], 1u);
        atomicAdd(&s_hist[((v2 - vb) << phsh) + phase], 1u); atomicAdd(&s_hist[((v3 - vb) << phsh) + phase], 1u);
      } else {
        value_hist_add(s_hist, ghist, aux, v0, phase, vb, width, phsh); value_hist_add(s_hist, ghist, aux, v1, phase, vb, width, phsh);
        value_hist_add(s_hist, ghist, aux, v2, phase, vb, width, phsh); value_hist_add(s_hist, ghist, aux, v3, phase, vb, width, phsh);
        const int h01 = v0 > v1 ? v0 : v1, h23 = v2 > v3 ? v2 : v3, h = h01 > h23 ? h01 : h23;
        lane_hi = h > lane_hi ? h : lane_hi;
      }
      if (out) *reinterpret_cast<int4*>(out + q) = make_int4(v0, v1, v2, v3);
      if (out8) {
        auto sat = [](int v) -> uint32_t { return v < 0 ? 0u : (v > kByteSat ? (uint32_t)kByteSat : (uint32_t)v); };
        *reinterpret_cast<uint32_t*>(out8 + q) = sat(v0) | (sat(v1) << 8) | (sat(v2) << 16) | (sat(v3) << 24);
      }
    }
  };
  TileRegs ra, rb;
  GcRegs ga = {0, 0}, gb = {0, 0};
  int64_t tile = blockIdx.x;
  if (tile < ntiles) {
    tile_request(ra, depth, tile * kTileBases, n);
    if (ADJUST) gc_tile_request(ga, gcbits, nwords, tile * kGcWords);
  }
  while (tile < ntiles) {
    trip(ra, ga, rb, gb, tile);
    tile += gridDim.x;
    if (tile >= ntiles) break;
    trip(rb, gb, ra, ga, tile);
    tile += gridDim.x;
  }
  __syncthreads();
  if (out8 && blockIdx.x == 0 && threadIdx.x < 4) {   // the ragged last n % 4 bases (the loop consumes whole quads)
    const int64_t i = (n & ~(int64_t)3) + threadIdx.x;
    if (i < n) { const int v = depth[i]; out8[i] = (uint8_t)(v < 0 ? 0 : (v > kByteSat ? kByteSat : v)); }
  }
  if (materialize) return;   // the tail quirks are applied to out[] by a launch of their own (k_gc_tail_fixup_out)
  publish_hist_hi(lane_hi, aux);
  value_hist_finish<ADJUST, true>(s_hist, depth, gcbits, n, table, out, ghist, aux, hist_slabs, gsum, per_group, counters, vm, head_src, head_dst, head_bytes, vb, width, phsh, pp, cap_mult);
}


// ------------------------------------------------------------------------------------------
// K3'  value histogram of the rescaled depth from the byte copy: wave-autonomous like K2 (sub-tiles of 1024 bases, 16
// consecutive bases = ONE 16-byte load per lane, GC words in a per-wave LDS slot, no workgroup barrier in the loop),
// nothing written per base.  A lane whose 16 bytes contain the escape code fetches its 16 values from the int32 array.
struct Sub8Regs { uint4 b; uint64_t gw; };
__device__ inline void sub8_request(Sub8Regs& r, const uint8_t* __restrict__ d8, const uint64_t* __restrict__ gcbits,
                                    int64_t nwords, int64_t base, int lane) {
  r.b = *reinterpret_cast<const uint4*>(d8 + base + 16 * (int64_t)lane);   // the copy is padded to whole sub-tiles
  const int64_t w = base / 64 - kGcLeft + lane;
  r.gw = gcbits[(lane < kSubLds && w >= 0 && w < nwords) ? w : 0];
}

__global__ __launch_bounds__(kThreads, 4) void k_value_hist8(const uint8_t* __restrict__ d8, const int32_t* __restrict__ depth,
                                                          const uint64_t* __restrict__ gcbits, int64_t n, int64_t nwords,
                                                          const double* __restrict__ table, uint32_t* __restrict__ ghist,
                                                          ValueHistAux* __restrict__ aux, unsigned int* __restrict__ hist_slabs,
                                                          unsigned int* __restrict__ gsum, int per_group, unsigned int* __restrict__ counters,
                                                          ValueMedian* __restrict__ vm, const void* head_src, void* head_dst,
                                                          unsigned int head_bytes, uint8_t* __restrict__ out8,
                                                          const unsigned int* __restrict__ escapes, unsigned int escape_limit) {
  __shared__ WaveGc s_gc[kThreads / 64];
  __shared__ double s_table[kGcLevels];
  __shared__ float s_ratio[kGcLevels];
  // [512 values][16 lane phases]: a GC level with an odd mean (a stretch of lower-case sequence next to N runs: table[0] =
  // 3.2 where the depth is 21) rescales its bases by 9 -- past 255, where every value used to be a global atomic on one of
  // a few hundred words: 0.2 % of a chromosome's bases cost the kernel 40 % more time.
  __shared__ unsigned int s_hist[kK3Width * kK3Phases];
  // Deep coverage: most bases did not fit K2's byte copy.  Nothing to do here but hand the header (with that count) to the
  // host, which sends the chromosome through the int32 kernels instead (per_base_phase).
  if (escapes[0] > escape_limit) {
    if (blockIdx.x == 0) export_words(head_dst, head_src, head_bytes);
    return;
  }
  const double rdmean = table[kGcLevels];
  for (int e = threadIdx.x; e < kK3Width * kK3Phases; e += kThreads) s_hist[e] = 0;
  for (int e = threadIdx.x; e < kGcLevels; e += kThreads) { const double t = table[e]; s_table[e] = t; s_ratio[e] = (float)(rdmean / t); }
  __syncthreads();
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  WaveGc& G = s_gc[wave];
  const int phase = lane & (kK3Phases - 1);
  int lane_hi = 0;   // largest value this lane sent past the LDS range
  const int64_t nsub = (n + kSubBases - 1) / kSubBases;
  const int64_t stride = (int64_t)gridDim.x * (kThreads / 64);
  const int64_t whole = n & ~(int64_t)3;   // the ragged last n % 4 bases belong to the tail fixup, as in K3

  auto rescale = [&](int d, uint32_t g) { return (int)((double)d * rdmean / s_table[g] + 0.5); };   // gccontent.cpp:89, truncation
  auto trip = [&](const Sub8Regs& cur, Sub8Regs& nxt, int64_t sub) {
    const int64_t base = sub * kSubBases;
    const int64_t first_bit = base - kGcLeft * 64;
    {
      const int64_t w = base / 64 - kGcLeft + lane;
      const uint64_t word = (lane < kSubLds && w >= 0 && w < nwords) ? cur.gw : 0;
      if (lane < kSubLds + 1) G.word[lane] = word;
    }
    if (sub + stride < nsub) sub8_request(nxt, d8, gcbits, nwords, (sub + stride) * kSubBases, lane);
    __builtin_amdgcn_wave_barrier();
    const int64_t i0 = base + 16 * (int64_t)lane;
    const bool interior = base >= 101 && base + kSubBases - 1 <= n - 102 && base + kSubBases <= whole;
    if (interior) {
      const uint32_t rel = (uint32_t)(i0 - 100 - first_bit);
      uint32_t cnt = wgc_window(G, rel);
      const uint32_t leave = wgc_field16(G, rel), enter = wgc_field16(G, rel + 201);
      const uint32_t w4[4] = {cur.b.x, cur.b.y, cur.b.z, cur.b.w};
      const bool esc = has_escape(w4[0]) || has_escape(w4[1]) || has_escape(w4[2]) || has_escape(w4[3]);
      const uint32_t cnt0 = cnt;
      bool redo = esc;
      int v[16];
      if (!esc) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          v[j] = (int)rescale_f32((float)((w4[j >> 2] >> (8 * (j & 3))) & 0xffu), s_ratio[cnt], redo);
          cnt = cnt - ((leave >> j) & 1u) + ((enter >> j) & 1u);
        }
      }
      if (redo) {   // rare: an escape byte (the lane's values come from the int32 array) or a rescaled value too close to
                    // an integer boundary for the ratio form: the reference's own expression for the lane's sixteen bases
        cnt = cnt0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int4 x;
          if (esc) x = *reinterpret_cast<const int4*>(depth + i0 + 4 * q);
          else x = make_int4((int)(w4[q] & 0xffu), (int)((w4[q] >> 8) & 0xffu), (int)((w4[q] >> 16) & 0xffu), (int)(w4[q] >> 24));
          v[4 * q] = rescale(x.x, cnt); cnt = cnt - ((leave >> (4 * q)) & 1u) + ((enter >> (4 * q)) & 1u);
          v[4 * q + 1] = rescale(x.y, cnt); cnt = cnt - ((leave >> (4 * q + 1)) & 1u) + ((enter >> (4 * q + 1)) & 1u);
          v[4 * q + 2] = rescale(x.z, cnt); cnt = cnt - ((leave >> (4 * q + 2)) & 1u) + ((enter >> (4 * q + 2)) & 1u);
          v[4 * q + 3] = rescale(x.w, cnt); cnt = cnt - ((leave >> (4 * q + 3)) & 1u) + ((enter >> (4 * q + 3)) & 1u);
        }
      }
      unsigned ored = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) ored |= (unsigned)v[j];
      {   // the rescaled values as bytes, saturated at kByteSat ("this much or more"), for the kernel that caps them anyway
        uint32_t pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int a = v[4 * q] > kByteSat ? kByteSat : v[4 * q], b = v[4 * q + 1] > kByteSat ? kByteSat : v[4 * q + 1];
          const int c = v[4 * q + 2] > kByteSat ? kByteSat : v[4 * q + 2], d = v[4 * q + 3] > kByteSat ? kByteSat : v[4 * q + 3];
          pk[q] = (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
        }
        *reinterpret_cast<uint4*>(out8 + i0) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      }
      if (ored < (unsigned)kK3Width) {   // the common case, branch-free: sixteen LDS atomics into [value][lane phase]
#pragma unroll
        for (int j = 0; j < 16; ++j) atomicAdd(&s_hist[v[j] * kK3Phases + phase], 1u);
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) { value_hist_add(s_hist, ghist, aux, v[j], phase, 0, kK3Width, kK3PhaseShift); lane_hi = v[j] > lane_hi ? v[j] : lane_hi; }
      }
    } else {          // edge sub-tiles: the reference's clamped windows (App. A Q1), whole quads only, straight from the int32 array
      for (int j = 0; j < 16; ++j) {
        const int64_t i = i0 + j;
        if (i >= whole) break;
        int64_t lo = i - 100;
        if (lo < 0) lo = 0;
        if (lo > n - 202) lo = n - 202;
        const uint32_t rel = (uint32_t)(lo - first_bit);
        const int ve = rescale(depth[i], wgc_window(G, rel));
        value_hist_add(s_hist, ghist, aux, ve, phase, 0, kK3Width, kK3PhaseShift);
        lane_hi = ve > lane_hi ? ve : lane_hi;
        out8[i] = (uint8_t)(ve > kByteSat ? kByteSat : ve);
      }
    }
    __builtin_amdgcn_wave_barrier();   // the slot is rewritten by the next trip
  };

  Sub8Regs ra, rb;
  int64_t sub = (int64_t)blockIdx.x * (kThreads / 64) + wave;
  if (sub < nsub) sub8_request(ra, d8, gcbits, nwords, sub * kSubBases, lane);
  while (sub < nsub) {
    trip(ra, rb, sub);
    sub += stride;
    if (sub >= nsub) break;
    trip(rb, ra, sub);
    sub += stride;
  }
  __syncthreads();
  publish_hist_hi(lane_hi, aux);
  value_hist_finish<true>(s_hist, depth, gcbits, n, table, nullptr, ghist, aux, hist_slabs, gsum, per_group, counters, vm, head_src, head_dst, head_bytes, 0, kK3Width, kK3PhaseShift);
}

// ------------------------------------------------------------------------------------------
// K4  cap_compact_bin.  One workgroup per tile of TB bins (TB*m compacted bases) staged in LDS.
//
// Compaction is a piecewise shift: compacted index p maps to source index p + cum[k], k = number of
// removed regions with cbreak <= p.  A tile that no region cuts (nearly all of them) is one
// contiguous source range; its 16-byte loads (destination-aligned, so the compacted copy leaves as
// 16-byte stores straight from registers) are requested one tile ahead of use, i.e. the loads of
// tile t+1 are in flight while the medians of tile t are computed.
// Tiles cut by a region take the generic segment loop.
//
// Per value: cap, LDS store, one LDS atomic into the [value][MAD residue class] histogram.
// Sum / sum of squares / median of the chromosome are all derived from that histogram on the host.
struct __attribute__((packed, aligned(4))) Quad4 { int x, y, z, w; };   // 16 bytes, dword-aligned

// rare path (values outside the LDS range, partial quads): kept out of line
__device__ __attribute__((noinline)) void hist_value(unsigned int* s_hist, uint32_t* __restrict__ res_hist, BinAccum* acc, int vr, int vb, int x, int cls, int pack16) {
  if (x >= vb && x - vb < vr) { const int idx = (x - vb) * kResClasses + cls; if (pack16) atomicAdd(&s_hist[idx >> 1], 1u << ((idx & 1) << 4)); else atomicAdd(&s_hist[idx], 1u); }
  else if (x >= 0 && x < kHistValues) atomicAdd(&res_hist[(size_t)x * kResClasses + cls], 1u);
  else { atomicAdd(&acc->big, 1ull); atomicMax(&acc->vmax, (unsigned int)x); }
}

// MAXV: 16-byte loads per thread and tile.  EPT: values per thread in the median phase held in
// registers (4 threads per bin); EPT = 0 keeps them in LDS (any m / any thread split).
template <int MAXV, int EPT>
__global__ __launch_bounds__(kThreads, (MAXV <= 8 ? 3 : 1)) void k_cap_compact_bin(
    const int32_t* __restrict__ src, int64_t n, const int64_t* __restrict__ cbreak, const int64_t* __restrict__ cum,
    int nreg, int64_t ncompact, int32_t capval, int m, int TB, int vr /* LDS histogram value range, power of two */,
    int vb /* first value of the LDS histogram's window: 0 unless the coverage is deep (k4_window_base) */, int32_t* __restrict__ rdc, int32_t* __restrict__ binmed, int64_t* __restrict__ binsum,
    uint32_t* __restrict__ res_hist, BinAccum* __restrict__ acc, unsigned int* __restrict__ hist_slabs,
    unsigned int* __restrict__ gsum, int per_group, unsigned int* __restrict__ counters, int overwrite,
    const void* exp_src, void* exp_dst, unsigned int exp_bytes, K4Regions inl,
    int pack16 /* the LDS histogram as 16-bit counters, two to a word (deep coverage: a 512-value window in 32 KB instead of 64 -- two
                  workgroups per CU instead of one; the caller guarantees fewer than 65536 x 31 values per workgroup); the
                  workgroups' slabs are unpacked, their sums go straight into res_hist (cleared by the caller) */) {
  extern __shared__ __align__(16) unsigned char smem[];
  int32_t* s_val = reinterpret_cast<int32_t*>(smem);                       // TB*m values (padded to 4)
  const int tile_elems = TB * m;
  const int tile_pad = (tile_elems + 3) & ~3;
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + (size_t)tile_pad * 4);   // [vr][32]
  __shared__ int64_t s_break[kRegLds], s_cum[kRegLds + 1];
  for (int e = threadIdx.x; e < (pack16 ? vr * kResClasses / 2 : vr * kResClasses); e += kThreads) s_hist[e] = 0;
  if (nreg <= kRegInline) {   // the short list travels with the kernel arguments: no upload in front of the launch
    for (int e = threadIdx.x; e < nreg; e += kThreads) s_break[e] = inl.brk[e];
    for (int e = threadIdx.x; e <= nreg; e += kThreads) s_cum[e] = inl.cum[e];
  } else {
    for (int e = threadIdx.x; e < kRegLds && e < nreg; e += kThreads) s_break[e] = cbreak[e];
    for (int e = threadIdx.x; e <= kRegLds && e <= nreg; e += kThreads) s_cum[e] = cum[e];
  }
  __syncthreads();
  const RegionTable R{cbreak, cum, nreg, s_break, s_cum};

  const int64_t lim31 = (ncompact / 31) * 31;
  const int64_t nb = ncompact / m;
  const int64_t ntiles = (ncompact + tile_elems - 1) / tile_elems;
  const int parts = kThreads / TB;          // threads cooperating on one bin
  const int kth = (m + 1) / 2;              // rank of the median, m odd (rsi.cpp:2061)
  const bool sw16 = capval >= 0 && capval < 32767 && (int64_t)m * capval < (int64_t)1 << 31;   // every value of the tile fits a 16-bit field with a spare bit

  // tile geometry: k = regions cut out at or before P0; a tile is `plain` when no region cuts it
  int k = 0;
  auto geometry = [&](int64_t tile, int64_t& P0, int64_t& P1, bool& plain, int64_t& soff) {
    P0 = tile * tile_elems;
    P1 = (P0 + tile_elems < ncompact) ? P0 + tile_elems : ncompact;
    while (k < nreg && R.brk(k) <= P0) ++k;   // tiles are visited in increasing order
    plain = (k >= nreg) || (R.brk(k) >= P1);
    soff = P0 + R.shift(k);
  };
  // Destination-aligned quads: thread idx owns compacted elements 4*idx .. 4*idx+3 of the tile and
  // loads them with one 16-byte load from src + soff + 4*idx, which is only dword-aligned in the
  // source (global_load_dwordx4 needs no more on gfx950).  Whole quads only; the partial quad at
  // the end of the chromosome's last tile is fetched at consume time.
  int4 regs[MAXV];
  auto request = [&](int64_t soff, int cnt) {   // branch-free: quads outside the tile re-read quad 0 (ignored later)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int e0 = 4 * (i * kThreads + (int)threadIdx.x);
      const Quad4 v = *reinterpret_cast<const Quad4*>(src + (e0 + 4 <= cnt ? soff + e0 : 0));
      regs[i] = make_int4(v.x, v.y, v.z, v.w);
    }
  };

  int64_t P0, P1, soff; bool plain;
  int64_t tile = blockIdx.x;
  if (tile < ntiles) { geometry(tile, P0, P1, plain, soff); if (plain) request(soff, (int)(P1 - P0)); }
  for (; tile < ntiles; tile += gridDim.x) {
    const int cnt = (int)(P1 - P0);
    __syncthreads();   // s_val free (and s_hist zeroed on the first trip)
    if (plain) {
      // ---- consume the requested quads: cap, LDS store (16 bytes), histogram, rdc store (16 bytes) ----
      const uint32_t p0mod = (uint32_t)(P0 % 31);
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int e0 = 4 * (i * kThreads + (int)threadIdx.x);   // tile-local index of the quad's first value
        if (e0 >= cnt) continue;
        int4 d = regs[i];
        const bool whole = e0 + 4 <= cnt;
        if (!whole) {   // the chromosome's last, partial quad
          d.x = src[soff + e0]; d.y = e0 + 1 < cnt ? src[soff + e0 + 1] : 0; d.z = e0 + 2 < cnt ? src[soff + e0 + 2] : 0; d.w = 0;
        }
        d.x = d.x > capval ? capval : d.x; d.y = d.y > capval ? capval : d.y;
        d.z = d.z > capval ? capval : d.z; d.w = d.w > capval ? capval : d.w;
        *reinterpret_cast<int4*>(s_val + e0) = d;
        if (whole) *reinterpret_cast<int4*>(rdc + P0 + e0) = d;
        else { rdc[P0 + e0] = d.x; if (e0 + 1 < cnt) rdc[P0 + e0 + 1] = d.y; if (e0 + 2 < cnt) rdc[P0 + e0 + 2] = d.z; }
        uint32_t c0 = (p0mod + (uint32_t)e0) % 31u;            // (P0 + e0) mod 31
        const uint32_t c1 = c0 == 30u ? 0u : c0 + 1u, c2 = c1 == 30u ? 0u : c1 + 1u, c3 = c2 == 30u ? 0u : c2 + 1u;
        const int wx = d.x - vb, wy = d.y - vb, wz = d.z - vb, ww = d.w - vb;   // positions in the histogram's window
        if (whole && P0 + e0 + 3 < lim31 && ((unsigned)wx | (unsigned)wy | (unsigned)wz | (unsigned)ww) < (unsigned)vr) {
          // the common case, branch-free: four LDS atomics into [value][residue class]
          if (pack16) {
            const int i0 = wx * kResClasses + (int)c0, i1 = wy * kResClasses + (int)c1, i2 = wz * kResClasses + (int)c2, i3 = ww * kResClasses + (int)c3;
            atomicAdd(&s_hist[i0 >> 1], 1u << ((i0 & 1) << 4)); atomicAdd(&s_hist[i1 >> 1], 1u << ((i1 & 1) << 4));
            atomicAdd(&s_hist[i2 >> 1], 1u << ((i2 & 1) << 4)); atomicAdd(&s_hist[i3 >> 1], 1u << ((i3 & 1) << 4));
          } else {
            atomicAdd(&s_hist[wx * kResClasses + c0], 1u); atomicAdd(&s_hist[wy * kResClasses + c1], 1u);
            atomicAdd(&s_hist[wz * kResClasses + c2], 1u); atomicAdd(&s_hist[ww * kResClasses + c3], 1u);
          }
        } else {
          auto count = [&](int x, int e, uint32_t cls) {
            if (e < cnt) hist_value(s_hist, res_hist, acc, vr, vb, x, (P0 + e) < lim31 ? (int)cls : 31, pack16);
          };
          count(d.x, e0, c0); count(d.y, e0 + 1, c1); count(d.z, e0 + 2, c2); count(d.w, e0 + 3, c3);
        }
      }
    } else {
      // ---- generic: contiguous source segments between removed regions ----
      int kk = k;
      int64_t seg = P0;
      while (seg < P1) {
        const int64_t nxt = (kk < nreg && R.brk(kk) < P1) ? R.brk(kk) : P1;
        const int64_t len = nxt - seg;
        if (len > 0) {
          const int64_t so = seg + R.shift(kk);
          const int dst = (int)(seg - P0);
          for (int64_t e = threadIdx.x; e < len; e += kThreads) {
            int x = src[so + e];
            if (x > capval) x = capval;
            s_val[dst + e] = x;
            rdc[seg + e] = x;
            const int64_t p = seg + e;
            hist_value(s_hist, res_hist, acc, vr, vb, x, p < lim31 ? (int)((uint32_t)p % 31u) : 31, pack16);
          }
        }
        seg = nxt;
        if (kk < nreg && R.brk(kk) == nxt) ++kk;
      }
    }
    __syncthreads();
    // ---- request the next tile now: its loads fly during the store + median phase ----
    const int64_t cur_tile = tile;
    if (tile + gridDim.x < ntiles) { geometry(tile + gridDim.x, P0, P1, plain, soff); if (plain) request(soff, (int)(P1 - P0)); }
    // ---- per-bin exact median (order statistic kth) and sum: `parts` threads per bin ----
    const int b_local = threadIdx.x / parts, part = threadIdx.x % parts;
    const int64_t b = cur_tile * TB + b_local;
    const bool active = b < nb;
    const int32_t* x = s_val + b_local * m;
    int lo = 0x7fffffff, hi = (int)0x80000000;
    long long ssum = 0;
    if (EPT > 0 && sw16) {
      // Values below 2^15 (a cap is in force and lies below): two to a register as 16-bit fields, bit 15 of each as the guard --
      // #{x > t} of two values is one subtraction and one popcount, as in the byte kernels' median phase; slots past the bin
      // hold 0xffff, above every t.  The median lies next to the bin's mean: 64 values around it bracket it on all but a
      // handful of bins (the two counts that prove it, then six bisection steps); a wave with a bin outside its bracket
      // bisects [0, cap].  32-bit sums: m * cap stays below 2^31.  (The int32 form below cost 40 of this kernel's 80 vector
      // instructions per base at 300x: up to ten data-dependent steps of 26 compares over min .. max of the bin.)
      constexpr int kPairs = (EPT + 1) / 2;
      uint32_t pk[EPT > 0 ? kPairs : 1];
      uint32_t s32 = 0;
#pragma unroll
      for (int i = 0; i < kPairs; ++i) {
        const int j0 = part + parts * (2 * i), j1 = part + parts * (2 * i + 1);
        const bool in0 = active && j0 < m, in1 = 2 * i + 1 < EPT && active && j1 < m;
        const uint32_t a = in0 ? (uint32_t)x[j0] : 0xffffu, c = in1 ? (uint32_t)x[j1] : 0xffffu;
        s32 += (in0 ? a : 0u) + (in1 ? c : 0u);
        pk[i] = a | (c << 16) | 0x80008000u;
      }
      s32 = (uint32_t)parts_sum((int)s32, parts);
      auto count_le = [&](int t) {   // #{x <= t} of the bin, t = -1 .. 32766
        const uint32_t sub = (uint32_t)(t + 1) * 0x00010001u;
        int gt = 0;
#pragma unroll
        for (int i = 0; i < kPairs; ++i) gt += __popc((pk[i] - sub) & 0x80008000u);
        return 2 * kPairs * parts - parts_sum(gt, parts);
      };
      int blo = 0, bhi = capval, steps = 15;
      {
        const int est = (int)((float)s32 / (float)m);
        int lo0 = est - 31;
        lo0 = lo0 < 0 ? 0 : lo0;
        int hi0 = lo0 + 63;
        hi0 = hi0 > capval ? capval : hi0;
        lo0 = lo0 > hi0 ? hi0 : lo0;
        const bool below = count_le(lo0 - 1) < kth;
        const bool above = hi0 >= capval || count_le(hi0) >= kth;
        if (__all((below && above) || !active)) { blo = lo0; bhi = hi0; steps = 6; }
      }
#pragma unroll 1
      for (int it = 0; it < steps; ++it) {
        const int mid = (blo + bhi) >> 1;
        const int le = count_le(mid);
        if (blo < bhi) { if (le >= kth) bhi = mid; else blo = mid + 1; }
      }
      lo = blo; ssum = (long long)s32;
    } else if (EPT > 0) {
      // the thread's share of the bin in registers; slots past the bin hold INT_MAX (never <= mid)
      int r[EPT > 0 ? EPT : 1];
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const int j = part + parts * i;
        r[i] = (active && j < m) ? x[j] : 0x7fffffff;
      }
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        const bool in = active && part + parts * i < m;
        const int v = r[i];
        lo = (in && v < lo) ? v : lo; hi = (in && v > hi) ? v : hi; ssum += in ? v : 0;
      }
      for (int d = 1; d < parts; d <<= 1) {
        const int olo = __shfl_xor(lo, d), ohi = __shfl_xor(hi, d);
        const long long os = __shfl_xor(ssum, d);
        lo = olo < lo ? olo : lo; hi = ohi > hi ? ohi : hi; ssum += os;
      }
      while (__any(active && lo < hi)) {   // bisection on the value: smallest v with #{x <= v} >= kth
        const int mid = (int)(((long long)lo + (long long)hi) >> 1);
        int c = 0;
#pragma unroll
        for (int i = 0; i < EPT; ++i) c += r[i] <= mid;
        c = parts_sum(c, parts);
        if (active && lo < hi) { if (c >= kth) hi = mid; else lo = mid + 1; }
      }
    } else {
      if (active) for (int j = part; j < m; j += parts) { const int v = x[j]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; ssum += v; }
      for (int d = 1; d < parts; d <<= 1) {
        const int olo = __shfl_xor(lo, d), ohi = __shfl_xor(hi, d);
        const long long os = __shfl_xor(ssum, d);
        lo = olo < lo ? olo : lo; hi = ohi > hi ? ohi : hi; ssum += os;
      }
      while (__any(active && lo < hi)) {
        const int mid = (int)(((long long)lo + (long long)hi) >> 1);
        int c = 0;
        if (active && lo < hi) for (int j = part; j < m; j += parts) c += x[j] <= mid;
        c = parts_sum(c, parts);
        if (active && lo < hi) { if (c >= kth) hi = mid; else lo = mid + 1; }
      }
    }
    if (active && part == 0) { binmed[b] = lo; binsum[b] = ssum; }
  }
  __syncthreads();
  // ---- the LDS histogram leaves as this workgroup's slab (plain coalesced stores); k_hist_slab_reduce
  // folds the slabs into res_hist.  Atomics from every workgroup into the same few thousand words
  // would serialise on them. ----
  unsigned int* slab = hist_slabs + (size_t)blockIdx.x * vr * kResClasses;
  if (pack16) {
    for (int e = threadIdx.x; e < vr * kResClasses; e += kThreads) st_cg(&slab[e], (s_hist[e >> 1] >> ((e & 1) << 4)) & 0xffffu);
    // the groups' sums go straight into res_hist's window (the caller cleared res_hist): no 64 KB total in the last workgroup's LDS
    if (!fold_slabs_add(hist_slabs, res_hist + (size_t)vb * kResClasses, vr * kResClasses, per_group, counters)) return;
    export_words(exp_dst, exp_src, exp_bytes);
    return;
  }
  for (int e = threadIdx.x; e < vr * kResClasses; e += kThreads) st_cg(&slab[e], s_hist[e]);
  // ---- the last workgroup to finish folds the slabs into res_hist (device_util.h) and hands the histogram, with the
  // BinAccum record in front of it, to the host through mapped memory: no fold launch, no device -> host copy.
  // overwrite: every value is below vr (the cap is), so res_hist needs no clearing beforehand ----
  if (!fold_slabs(hist_slabs, gsum, s_hist, vr * kResClasses, per_group, counters)) return;
  for (int e = threadIdx.x; e < vr * kResClasses; e += kThreads) {
    const unsigned int c = s_hist[e];
    if (overwrite) st_cg(&res_hist[e], c); else if (c) atomicAdd(&res_hist[(size_t)vb * kResClasses + e], c);
  }
  sync_drained();
  export_words(exp_dst, exp_src, exp_bytes);
}


// ------------------------------------------------------------------------------------------
// K4'  cap_compact_bin8: K4 fed from the byte copy of the GC-RESCALED depth that K3' leaves behind (values saturated at
// kByteSat: the kernel is used when the cap is lower, so a saturated value is capped either way).  The rescaled int32 array is
// never written or read: per base this kernel reads 1 byte and writes the 4 bytes of the capped, compacted depth -- 5 B/base
// against 8.4 for K4 behind a K3 that writes 4 more -- and its per-base work is unpack, cap, one LDS atomic, pack.
//
// A plain tile (no removed region cuts it) is a contiguous SOURCE range.  Chunks of 16 values are aligned in the COMPACTED
// array (64 aligned bytes of rdc, 16 aligned bytes of the LDS tile); in the source they start at any byte, which a 16-byte
// load does not mind; thread t owns chunk t (and t + 256), requested one tile ahead.  The tile's values live in LDS as BYTES
// (ds_write_b128, conflict-free) and the per-bin median phase works on them four to a register.  Tiles that touch the
// chromosome's ends (clamped windows, App. A Q1; the tail quirks of the 20-slice write-back, Q2/Q3) or are cut by a removed
// region take a per-element path that recomputes the rescale from the int32 array.

template <int MAXC, int EPT, bool SW7>
__global__ __launch_bounds__(kThreads, 4) void k_cap_compact_bin8(
    const uint8_t* __restrict__ r8 /* rescaled, saturated bytes */, const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits,
    int64_t n, int64_t nwords, const double* __restrict__ table /* [kGcLevels] + rdmean */, const int64_t* __restrict__ cbreak,
    const int64_t* __restrict__ cum, int nreg, int64_t ncompact, int32_t capval, int m, int TB, int vr, uint8_t* __restrict__ rdc8 /* capped + compacted depth, one byte per base */,
    int32_t* __restrict__ binmed,
    int64_t* __restrict__ binsum, uint32_t* __restrict__ res_hist, unsigned int* __restrict__ hist_slabs, unsigned int* __restrict__ gsum,
    int per_group, unsigned int* __restrict__ counters, const void* exp_src, void* exp_dst, unsigned int exp_bytes, K4Regions inl,
    int raw /* 1: r8 is the RAW depth (-NOGC): no rescale and no slice quirks on the per-element path either */) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* s_val = smem;                                                              // MAXC * 256 chunks of 16 bytes
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + (size_t)MAXC * kThreads * 16);   // [vr][32]
  __shared__ double s_table[kGcLevels];   // the per-element path's rescale
  __shared__ int64_t s_break[kRegLds], s_cum[kRegLds + 1];
  for (int e = threadIdx.x; e < vr * kResClasses; e += kThreads) s_hist[e] = 0;
  for (int e = threadIdx.x; e < kGcLevels; e += kThreads) s_table[e] = table[e];
  if (nreg <= kRegInline) {
    for (int e = threadIdx.x; e < nreg; e += kThreads) s_break[e] = inl.brk[e];
    for (int e = threadIdx.x; e <= nreg; e += kThreads) s_cum[e] = inl.cum[e];
  } else {
    for (int e = threadIdx.x; e < kRegLds && e < nreg; e += kThreads) s_break[e] = cbreak[e];
    for (int e = threadIdx.x; e <= kRegLds && e <= nreg; e += kThreads) s_cum[e] = cum[e];
  }
  const double rdmean = table[kGcLevels];
  __syncthreads();
  const RegionTable R{cbreak, cum, nreg, s_break, s_cum};

  const int64_t lim31 = (ncompact / 31) * 31;
  const int64_t nb = ncompact / m;
  const int tile_elems = TB * m;            // a multiple of 16 (TB = 64)
  const int nchunks = tile_elems / 16;
  const int64_t ntiles = (ncompact + tile_elems - 1) / tile_elems;
  const int parts = kThreads / TB;          // threads cooperating on one bin
  const int kth = (m + 1) / 2;              // rank of the median, m odd (rsi.cpp:2061)
  // the 20-slice write-back's tail (App. A Q2/Q3): cells n-201 .. n-201+r-1 carry the rescaled depth of the last r bases,
  // computed with the fresh edge window [n-201, n-1]; the last r bases keep their raw depth
  const int64_t S20 = n / 20, r20 = n - 20 * S20;
  const int64_t zone = n - 201;             // no fast-path tile may reach this base (also covers the clamped windows, i >= n-101)

  auto rescale = [&](int d, uint32_t g) { return (int)((double)d * rdmean / s_table[g] + 0.5); };   // gccontent.cpp:89, truncation
  auto slow_value = [&](int64_t i) -> int {   // the value K3 + its tail fixup would have left at source index i
    if (raw) return depth[i];
    if (r20 >= 2 && i >= n - 201 && i < n - 201 + r20) return rescale(depth[20 * S20 + (i - (n - 201))], (uint32_t)gc_count201(gcbits, n - 201));
    if (i >= 20 * S20) return depth[i];
    int64_t lo = i - 100;
    if (lo < 0) lo = 0;
    if (lo > n - 202) lo = n - 202;
    return rescale(depth[i], (uint32_t)gc_count201(gcbits, lo));
  };

  int k = 0;
  auto geometry = [&](int64_t tile, int64_t& P0, int64_t& P1, bool& fast, int64_t& soff) {
    P0 = tile * tile_elems;
    P1 = (P0 + tile_elems < ncompact) ? P0 + tile_elems : ncompact;
    while (k < nreg && R.brk(k) <= P0) ++k;   // tiles are visited in increasing order
    const bool plain = (k >= nreg) || (R.brk(k) >= P1);
    soff = P0 + R.shift(k);
    // fast: a whole tile, contiguous in the source, every base with an unclamped window, before the tail zone and before
    // the last partial stride of the 31 MAD residue classes
    fast = plain && P1 - P0 == tile_elems && P1 <= lim31 && soff >= 101 && soff + tile_elems <= zone;
  };
  // Chunks are aligned in the COMPACTED array (16 values = 64 aligned bytes of rdc, 16 aligned bytes of the LDS tile); in the
  // source they start at any byte, which a 16-byte load of the byte copy does not mind.
  uint4 regs[MAXC];
  auto request = [&](int64_t soff) {   // branch-free: chunks beyond the tile re-read chunk 0 (ignored later)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int kc = c * kThreads + (int)threadIdx.x;
      const Bytes16 b = *reinterpret_cast<const Bytes16*>(r8 + soff + 16 * (kc < nchunks ? kc : 0));
      regs[c] = make_uint4(b.x, b.y, b.z, b.w);
    }
  };

  int64_t P0, P1, soff; bool fast;
  int64_t tile = blockIdx.x;
  if (tile < ntiles) { geometry(tile, P0, P1, fast, soff); if (fast) request(soff); }
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();   // s_val is free (and s_hist zeroed on the first trip)
    if (fast) {
      const uint32_t p0mod = (uint32_t)(P0 % 31);
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int kc = c * kThreads + (int)threadIdx.x;
        if (kc >= nchunks) continue;
        const uint32_t w4[4] = {regs[c].x, regs[c].y, regs[c].z, regs[c].w};
        int v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) { const int x = (int)((w4[j >> 2] >> (8 * (j & 3))) & 0xffu); v[j] = x > capval ? capval : x; }
        uint32_t pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pk[q] = (uint32_t)v[4 * q] | ((uint32_t)v[4 * q + 1] << 8) | ((uint32_t)v[4 * q + 2] << 16) | ((uint32_t)v[4 * q + 3] << 24);
        *reinterpret_cast<uint4*>(s_val + 16 * kc) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(rdc8 + P0 + 16 * (int64_t)kc) = make_uint4(pk[0], pk[1], pk[2], pk[3]);   // 16-byte aligned: P0 is a multiple of 64 m
        // sixteen LDS atomics into [value][MAD residue class]: the class of element j is cls0 + j, minus 31 from the lane's
        // wrap point on; the element index rides in the instruction's offset field
        const uint32_t cls0 = (p0mod + 16u * (uint32_t)kc) % 31u;
        unsigned int* ha = s_hist + cls0;
        unsigned int* hb = ha - 31;
        const int jw = 31 - (int)cls0;
#pragma unroll
        for (int j = 0; j < 16; ++j) atomicAdd((j >= jw ? hb : ha) + v[j] * kResClasses + j, 1u);
      }
    } else {
      // ---- per-element path: contiguous source segments between removed regions, values from the int32 array ----
      int kk = k;
      int64_t seg = P0;
      while (seg < P1) {
        const int64_t nxt = (kk < nreg && R.brk(kk) < P1) ? R.brk(kk) : P1;
        const int64_t len = nxt - seg;
        if (len > 0) {
          const int64_t so = seg + R.shift(kk);
          const int dst = (int)(seg - P0);
          for (int64_t e = threadIdx.x; e < len; e += kThreads) {
            int x = slow_value(so + e);
            if (x > capval) x = capval;
            if (x < 0) x = 0;   // negative depth is refused by the caller (K2's flag); keep the byte store in range
            s_val[dst + e] = (unsigned char)x;
            rdc8[seg + e] = (unsigned char)x;
            const int64_t p = seg + e;
            atomicAdd(&s_hist[x * kResClasses + (p < lim31 ? (int)((uint32_t)p % 31u) : 31)], 1u);
          }
        }
        seg = nxt;
        if (kk < nreg && R.brk(kk) == nxt) ++kk;
      }
    }
    __syncthreads();
    // ---- request the next tile now: its loads fly during the median phase ----
    const int64_t cur_tile = tile;
    if (tile + gridDim.x < ntiles) { geometry(tile + gridDim.x, P0, P1, fast, soff); if (fast) request(soff); }
    // ---- per-bin exact median (order statistic kth) and sum: `parts` threads per bin ----
    const int b_local = threadIdx.x / parts, part = threadIdx.x % parts;
    const int64_t b = cur_tile * TB + b_local;
    const bool active = b < nb;
    if (SW7) {
      // Values below 128 (the cap is): four to a register, straight from the LDS bytes.  A bin is the bytes [B, B + m) of the
      // tile; its (up to 27) dwords go round robin to the bin's four threads, bytes outside the bin masked -- to 0 for the
      // sum (v_sad_u8 adds four bytes in one instruction), to 0xff for the counts.  #{x > mid} of four values is one
      // subtraction and one popcount: with the top bit of every byte set, (x | 0x80) - (mid + 1) keeps that bit exactly
      // where x > mid, and no byte borrows from its neighbour.  Bisection from [0, cap]: the same seven steps for every bin.
      const int B = b_local * m;
      const int d0 = B >> 2, d1 = (B + m - 1) >> 2;
      const uint32_t* w = reinterpret_cast<const uint32_t*>(s_val);
      uint32_t xo[7];
      uint32_t ssum = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int d = d0 + part + parts * i;
        const uint32_t v = w[d <= d1 ? d : d1];
        const int lo_cut = B - 4 * d, hi_cut = 4 * d + 4 - (B + m);          // bytes of the dword before / after the bin
        uint32_t keep = 0xffffffffu;
        keep = lo_cut > 0 ? keep << (8 * lo_cut) : keep;
        keep = hi_cut > 0 ? keep & (0xffffffffu >> (8 * hi_cut)) : keep;
        keep = (d <= d1 && active) ? keep : 0u;
        ssum = __builtin_amdgcn_sad_u8(v & keep, 0u, ssum);
        xo[i] = (v | ~keep) | 0x80808080u;
      }
      ssum = (uint32_t)parts_sum((int)ssum, parts);
      int lo = 0, hi = capval;
#pragma unroll 1
      for (int it = 0; it < 7; ++it) {
        const int mid = (lo + hi) >> 1;
        const uint32_t sub = (uint32_t)(mid + 1) * 0x01010101u;
        int gt = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) gt += __popc((xo[i] - sub) & 0x80808080u);
        gt = parts_sum(gt, parts);
        // masked bytes (0xff) always count as "> mid" (mid <= 126): 28 dword slots x 4 bytes - m of them per bin
        const int le = 4 * 7 * parts - gt;
        if (lo < hi) { if (le >= kth) hi = mid; else lo = mid + 1; }
      }
      if (active && part == 0) { binmed[b] = lo; binsum[b] = (int64_t)ssum; }
    } else if (EPT == 0) {
      // Caps of 128 .. 253 (a byte has no spare bit): the same scheme on 16-bit fields, two values to a register -- bytes 0
      // and 2 of a dword in one, bytes 1 and 3 in another, bit 15 of every field as the guard.  #{x > mid} of four values is
      // two subtractions and two popcounts; eight bisection steps from [0, cap].
      const int B = b_local * m;
      const int d0 = B >> 2, d1 = (B + m - 1) >> 2;
      const uint32_t* w = reinterpret_cast<const uint32_t*>(s_val);
      uint32_t xa[7], xc[7];
      uint32_t ssum = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int d = d0 + part + parts * i;
        const uint32_t v = w[d <= d1 ? d : d1];
        const int lo_cut = B - 4 * d, hi_cut = 4 * d + 4 - (B + m);          // bytes of the dword before / after the bin
        uint32_t keep = 0xffffffffu;
        keep = lo_cut > 0 ? keep << (8 * lo_cut) : keep;
        keep = hi_cut > 0 ? keep & (0xffffffffu >> (8 * hi_cut)) : keep;
        keep = (d <= d1 && active) ? keep : 0u;
        ssum = __builtin_amdgcn_sad_u8(v & keep, 0u, ssum);
        const uint32_t xb = v | ~keep;                                       // bytes outside the bin: 0xff, above every mid
        xa[i] = (xb & 0x00ff00ffu) | 0x80008000u;
        xc[i] = ((xb >> 8) & 0x00ff00ffu) | 0x80008000u;
      }
      ssum = (uint32_t)parts_sum((int)ssum, parts);
      int lo = 0, hi = capval;
#pragma unroll 1
      for (int it = 0; it < 8; ++it) {
        const int mid = (lo + hi) >> 1;
        const uint32_t sub = (uint32_t)(mid + 1) * 0x00010001u;
        int gt = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) gt += __popc((xa[i] - sub) & 0x80008000u) + __popc((xc[i] - sub) & 0x80008000u);
        gt = parts_sum(gt, parts);
        const int le = 4 * 7 * parts - gt;   // masked bytes (0xff) always count as "> mid" (mid <= 252)
        if (lo < hi) { if (le >= kth) hi = mid; else lo = mid + 1; }
      }
      if (active && part == 0) { binmed[b] = lo; binsum[b] = (int64_t)ssum; }
    }
  }
  __syncthreads();
  // ---- per-workgroup histogram slab; the last workgroup folds them into res_hist (every value is below vr: overwrite)
  // and hands [BinAccum | histogram] to the host ----
  unsigned int* slab = hist_slabs + (size_t)blockIdx.x * vr * kResClasses;
  for (int e = threadIdx.x; e < vr * kResClasses; e += kThreads) st_cg(&slab[e], s_hist[e]);
  // res_hist is zero when the launch begins (K1's FillList): the groups' sums go straight into it
  if (!fold_slabs_add(hist_slabs, res_hist, vr * kResClasses, per_group, counters)) return;
  export_words(exp_dst, exp_src, exp_bytes);
}

// ------------------------------------------------------------------------------------------
// K4w  cap_compact_bin16: K4 for caps of 254 .. 32766 (deep coverage, or a generous cap) in the shape of the byte kernels.
// The source is the int32 array K4 would read (the GC-rescaled depth, or the raw depth under -NOGC); the tile lives in LDS as
// 16-bit values, the median phase works on them two to a register (bit 15 of each field as the guard), the [value][MAD residue
// class] histogram is a window of kK4Window values as 16-bit counters (32 KB; three workgroups per CU with the 16 KB tile) that
// follows the depth (hist_window_base); values outside the window go to res_hist as global atomics, inline.  Output: the
// capped, compacted depth as int32 (what every later stage of this envelope reads), bin medians and sums, res_hist.
// The int32 kernel spent 640 us per 60 Mb at 300x on the same work (quads with a per-quad branch and an out-of-line slow
// path that nearly every wave entered, two workgroups per CU, ten-step bisections over min .. max of each bin).
constexpr int kW16 = 14;   // dwords (pairs of values) per thread in the median phase: 28 values x `parts` threads >= m + 1
template <int MAXC>
__global__ __launch_bounds__(kThreads, 3) void k_cap_compact_bin16(
    const int32_t* __restrict__ src, const int64_t* __restrict__ cbreak, const int64_t* __restrict__ cum, int nreg, int64_t ncompact,
    int32_t capval, int m, int TB, int vb /* first value of the LDS window */, int32_t* __restrict__ rdc, int32_t* __restrict__ binmed,
    int64_t* __restrict__ binsum, uint32_t* __restrict__ res_hist /* zero before the launch */, unsigned int* __restrict__ hist_slabs,
    int per_group, unsigned int* __restrict__ counters, const void* exp_src, void* exp_dst, unsigned int exp_bytes, K4Regions inl) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint16_t* s_val = reinterpret_cast<uint16_t*>(smem);                                             // MAXC * 256 chunks of 16 values
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + (size_t)MAXC * kThreads * 32);     // [kK4Window][32] 16-bit counters
  __shared__ int64_t s_break[kRegLds], s_cum[kRegLds + 1];
  __shared__ unsigned int s_extra[2 * kResClasses];   // rows of their own for the two values that pile up outside a window: 0 and the cap
  constexpr int vr = kK4Window;
  for (int e = threadIdx.x; e < vr * kResClasses / 2; e += kThreads) s_hist[e] = 0;
  if (threadIdx.x < 2 * kResClasses) s_extra[threadIdx.x] = 0u;
  if (nreg <= kRegInline) {
    for (int e = threadIdx.x; e < nreg; e += kThreads) s_break[e] = inl.brk[e];
    for (int e = threadIdx.x; e <= nreg; e += kThreads) s_cum[e] = inl.cum[e];
  } else {
    for (int e = threadIdx.x; e < kRegLds && e < nreg; e += kThreads) s_break[e] = cbreak[e];
    for (int e = threadIdx.x; e <= kRegLds && e <= nreg; e += kThreads) s_cum[e] = cum[e];
  }
  __syncthreads();
  const RegionTable R{cbreak, cum, nreg, s_break, s_cum};
  const int64_t lim31 = (ncompact / 31) * 31;
  const int64_t nb = ncompact / m;
  const int tile_elems = TB * m;            // a multiple of 16
  const int nchunks = tile_elems / 16;
  const int64_t ntiles = (ncompact + tile_elems - 1) / tile_elems;
  const int parts = kThreads / TB;
  const int kth = (m + 1) / 2;              // rank of the median, m odd (rsi.cpp:2061)

  // one value into the histogram: the LDS window's 16-bit counter, or res_hist itself (every value is in 0 .. cap < 2^15)
  auto count_value = [&](int x, int cls) {
    const unsigned int w = (unsigned int)(x - vb);
    if (w < (unsigned int)vr) { const unsigned int idx = w * kResClasses + (unsigned int)cls; atomicAdd(&s_hist[idx >> 1], 1u << ((idx & 1u) << 4)); }
    else if (x <= 0 || x == capval) atomicAdd(&s_extra[(x > 0 ? kResClasses : 0) + cls], 1u);   // uncovered stretches, capped pile-ups: thousands on one address
    else atomicAdd(&res_hist[(size_t)x * kResClasses + cls], 1u);
  };
  int k = 0;
  auto geometry = [&](int64_t tile, int64_t& P0, int64_t& P1, bool& fast, int64_t& soff) {
    P0 = tile * tile_elems;
    P1 = (P0 + tile_elems < ncompact) ? P0 + tile_elems : ncompact;
    while (k < nreg && R.brk(k) <= P0) ++k;   // tiles are visited in increasing order
    const bool plain = (k >= nreg) || (R.brk(k) >= P1);
    soff = P0 + R.shift(k);
    fast = plain && P1 - P0 == tile_elems && P1 <= lim31;   // a whole tile, contiguous in the source, before the last partial stride of the 31 classes
  };
  int4 regs[MAXC][4];
  auto request = [&](int64_t soff) {   // branch-free: chunks beyond the tile re-read chunk 0 (ignored later)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int kc = c * kThreads + (int)threadIdx.x;
      const int32_t* sp = src + soff + 16 * (kc < nchunks ? kc : 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const Quad4 v = *reinterpret_cast<const Quad4*>(sp + 4 * q); regs[c][q] = make_int4(v.x, v.y, v.z, v.w); }
    }
  };

  int64_t P0, P1, soff; bool fast;
  int64_t tile = blockIdx.x;
  if (tile < ntiles) { geometry(tile, P0, P1, fast, soff); if (fast) request(soff); }
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();   // s_val is free (and s_hist zeroed on the first trip)
    if (fast) {
      const uint32_t p0mod = (uint32_t)(P0 % 31);
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int kc = c * kThreads + (int)threadIdx.x;
        if (kc >= nchunks) continue;
        int v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int4 d = regs[c][q];
          v[4 * q] = d.x > capval ? capval : d.x; v[4 * q + 1] = d.y > capval ? capval : d.y;
          v[4 * q + 2] = d.z > capval ? capval : d.z; v[4 * q + 3] = d.w > capval ? capval : d.w;
        }
        uint32_t pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = ((uint32_t)v[2 * i] & 0xffffu) | ((uint32_t)v[2 * i + 1] << 16);
        *reinterpret_cast<uint4*>(s_val + 16 * kc) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(s_val + 16 * kc + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        int32_t* out = rdc + P0 + 16 * (int64_t)kc;   // 64-byte aligned: P0 is a multiple of 16
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<int4*>(out + 4 * q) = make_int4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        // sixteen atomics into [value][MAD residue class]: the class of element j is cls0 + j, minus 31 from the lane's wrap point on
        const int cls0 = (int)((p0mod + 16u * (uint32_t)kc) % 31u);
        const int jw = 31 - cls0;
#pragma unroll
        for (int j = 0; j < 16; ++j) count_value(v[j], cls0 + j - (j >= jw ? 31 : 0));
      }
    } else {
      // ---- per-element path: contiguous source segments between removed regions; the chromosome's last, partial tile ----
      int kk = k;
      int64_t seg = P0;
      while (seg < P1) {
        const int64_t nxt = (kk < nreg && R.brk(kk) < P1) ? R.brk(kk) : P1;
        const int64_t len = nxt - seg;
        if (len > 0) {
          const int64_t so = seg + R.shift(kk);
          const int dst = (int)(seg - P0);
          for (int64_t e = threadIdx.x; e < len; e += kThreads) {
            int x = src[so + e];
            if (x > capval) x = capval;
            s_val[dst + e] = (uint16_t)x;
            rdc[seg + e] = x;
            const int64_t p = seg + e;
            count_value(x, p < lim31 ? (int)((uint32_t)p % 31u) : 31);
          }
        }
        seg = nxt;
        if (kk < nreg && R.brk(kk) == nxt) ++kk;
      }
    }
    __syncthreads();
    // ---- request the next tile now: its loads fly during the median phase ----
    const int64_t cur_tile = tile;
    if (tile + gridDim.x < ntiles) { geometry(tile + gridDim.x, P0, P1, fast, soff); if (fast) request(soff); }
    // ---- per-bin exact median (order statistic kth) and sum: `parts` threads per bin, the bin's dwords round robin ----
    const int b_local = threadIdx.x / parts, part = threadIdx.x % parts;
    const int64_t b = cur_tile * TB + b_local;
    const bool active = b < nb;
    const int B = b_local * m;                             // the bin is the values [B, B + m) of the tile
    const int d0 = B >> 1, d1 = (B + m - 1) >> 1;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(s_val);
    uint32_t xo[kW16];
    uint32_t ssum = 0;
#pragma unroll
    for (int i = 0; i < kW16; ++i) {
      const int d = d0 + part + parts * i;
      const uint32_t val = w[d <= d1 ? d : d1];
      uint32_t keep = 0xffffffffu;
      keep = 2 * d < B ? 0xffff0000u : keep;               // the low half lies before the bin
      keep = 2 * d + 1 >= B + m ? keep & 0x0000ffffu : keep;   // the high half behind it
      keep = (d <= d1 && active) ? keep : 0u;
      const uint32_t in = val & keep;
      ssum += (in & 0xffffu) + (in >> 16);
      xo[i] = (val | ~keep) | 0x80008000u;                 // halves outside the bin: 0xffff, above every t
    }
    ssum = (uint32_t)parts_sum((int)ssum, parts);
    auto count_le = [&](int t) {   // #{x <= t} of the bin, t = -1 .. 32766
      const uint32_t sub = (uint32_t)(t + 1) * 0x00010001u;
      int gt = 0;
#pragma unroll
      for (int i = 0; i < kW16; ++i) gt += __popc((xo[i] - sub) & 0x80008000u);
      return 2 * kW16 * parts - parts_sum(gt, parts);
    };
    // 64 values around the bin's mean bracket its median on all but a handful of bins: the two counts that prove it, then six
    // bisection steps; a wave with a bin outside its bracket bisects [0, cap] (the result is the same either way)
    int lo = 0, hi = capval, steps = 15;
    {
      const int est = (int)((float)ssum / (float)m);
      int lo0 = est - 31;
      lo0 = lo0 < 0 ? 0 : lo0;
      int hi0 = lo0 + 63;
      hi0 = hi0 > capval ? capval : hi0;
      lo0 = lo0 > hi0 ? hi0 : lo0;
      const bool below = count_le(lo0 - 1) < kth;
      const bool above = hi0 >= capval || count_le(hi0) >= kth;
      if (__all((below && above) || !active)) { lo = lo0; hi = hi0; steps = 6; }
    }
#pragma unroll 1
    for (int it = 0; it < steps; ++it) {
      const int mid = (lo + hi) >> 1;
      const int le = count_le(mid);
      if (lo < hi) { if (le >= kth) hi = mid; else lo = mid + 1; }
    }
    if (active && part == 0) { binmed[b] = lo; binsum[b] = (int64_t)ssum; }
  }
  __syncthreads();
  // ---- the window leaves as this workgroup's slab, unpacked (four counters per 16-byte store); the groups' sums go straight into
  // res_hist's window ----
  unsigned int* slab = hist_slabs + (size_t)blockIdx.x * vr * kResClasses;
  for (int q = threadIdx.x; q < vr * kResClasses / 4; q += kThreads) {
    const unsigned int a = s_hist[2 * q], c = s_hist[2 * q + 1];
    u32x4 v4; v4.x = a & 0xffffu; v4.y = a >> 16; v4.z = c & 0xffffu; v4.w = c >> 16;
    st_cg_x4(slab + 4 * q, v4);
  }
  if (threadIdx.x < 2 * kResClasses) {
    const unsigned int c = s_extra[threadIdx.x];
    if (c) atomicAdd(&res_hist[(size_t)(threadIdx.x < kResClasses ? 0 : capval) * kResClasses + (threadIdx.x & (kResClasses - 1))], c);
  }
  if (!fold_slabs_add(hist_slabs, res_hist + (size_t)vb * kResClasses, vr * kResClasses, per_group, counters)) return;
  export_words(exp_dst, exp_src, exp_bytes);
}

// FIX: the rescale is (byte * R[level] + 2^15) >> 16 with K2j's per-level 16.16 ratios, each verified there against the
// reference's expression for every depth byte; a level that failed the check carries bit 31 and its lane takes the exact
// expression.  Without (a chromosome that went through K2 + K3'): the float form with its exactness margin (rescale_f32).
constexpr int kK4jCols = 48;   // columns of K4j's LDS histogram (see there)
template <int MAXC, int EPT, bool SW7, bool FIX>
__global__ __launch_bounds__(kThreads, 4) void k_rescale_compact_bin8(
    const uint8_t* __restrict__ d8 /* K2j's byte copy of the raw depth (kByteEscape = look at the int32 array) */, const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits,
    int64_t n, int64_t nwords, const double* __restrict__ table /* [kGcLevels] + rdmean */, const int64_t* __restrict__ cbreak,
    const int64_t* __restrict__ cum, int nreg, int64_t ncompact, int32_t capval, int m, int TB, int vr, uint8_t* __restrict__ rdc8 /* capped + compacted depth, one byte per base */,
    int32_t* __restrict__ binmed,
    int64_t* __restrict__ binsum, uint32_t* __restrict__ res_hist, unsigned int* __restrict__ hist_slabs, unsigned int* __restrict__ gsum,
    int per_group, unsigned int* __restrict__ counters, const void* exp_src, void* exp_dst, unsigned int exp_bytes, K4Regions inl,
    const unsigned int* __restrict__ rtab /* FIX: [kGcLevels] ratios from K2j */,
    PhaseParams* __restrict__ pp /* not NULL: regions, length and cap come from the device (K1b's and K2j's last workgroups), launched behind
                                   K2j without the host in between; the launch configuration (vr, SW7) was a guess to be checked here */) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* s_val = smem;                                                              // MAXC * 256 chunks of 16 bytes
  // [vr][kK4jCols]: columns 0 .. 30 the MAD residue classes, 31 .. 46 classes 0 .. 15 AGAIN (a lane's sixteen consecutive classes
  // then never wrap: no per-base select between two row pointers), 47 the class of the bases behind the last full stride of 31;
  // folded back to [vr][32] when the slab is written
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem + (size_t)MAXC * kThreads * 16);
  const int cols = vr <= 128 ? kK4jCols : kResClasses;   // a 256-value range keeps the plain 32 columns (48 KB otherwise: a workgroup per CU less)
  __shared__ unsigned int s_rt[kGcLevels];   // FIX: the levels' fixed-point ratios
  if (pp) {
    nreg = pp->nreg; ncompact = pp->ncompact; capval = pp->capval;
    // does the configuration this launch was given fit what the device found?  (wave-uniform: every workgroup decides alike)
    if (!pp->regions_ok || capval < 1 || capval >= kByteSat || capval >= vr || SW7 != (capval <= 127) || ncompact < (int64_t)m * 8) {
      if (blockIdx.x == 0 && threadIdx.x == 0) pp->redo = 1;
      return;
    }
  }
  __shared__ double s_table[kGcLevels];   // the reference's own expression: per-element path, escapes, values too close to a rounding boundary
  __shared__ float s_ratio[kGcLevels];    // rdmean / table[g] as float (rescale_f32)
  __shared__ uint64_t s_gw[kK4GcWords];   // GC mask words under the tile's source range (+ margins), staged per tile
  __shared__ int64_t s_break[kRegLds], s_cum[kRegLds + 1];
  for (int e = threadIdx.x; e < vr * cols; e += kThreads) s_hist[e] = 0;
  for (int e = threadIdx.x; e < kGcLevels; e += kThreads) { const double t = table[e]; s_table[e] = t; s_ratio[e] = (float)(table[kGcLevels] / t); }
  if (FIX) for (int e = threadIdx.x; e < kGcLevels; e += kThreads) s_rt[e] = rtab[e];
  if (nreg <= kRegInline && !pp) {
    for (int e = threadIdx.x; e < nreg; e += kThreads) s_break[e] = inl.brk[e];
    for (int e = threadIdx.x; e <= nreg; e += kThreads) s_cum[e] = inl.cum[e];
  } else {
    for (int e = threadIdx.x; e < kRegLds && e < nreg; e += kThreads) s_break[e] = cbreak[e];
    for (int e = threadIdx.x; e <= kRegLds && e <= nreg; e += kThreads) s_cum[e] = cum[e];
  }
  const double rdmean = table[kGcLevels];
  __syncthreads();
  const RegionTable R{cbreak, cum, nreg, s_break, s_cum};

  const int64_t lim31 = (ncompact / 31) * 31;
  const int64_t nb = ncompact / m;
  const int tile_elems = TB * m;            // a multiple of 16 (TB = 64)
  const int nchunks = tile_elems / 16;
  const int64_t ntiles = (ncompact + tile_elems - 1) / tile_elems;
  const int parts = kThreads / TB;          // threads cooperating on one bin
  const int kth = (m + 1) / 2;              // rank of the median, m odd (rsi.cpp:2061)
  const float inv_m = 1.0f / (float)m;      // for the median phase's first guess only (any guess gives the same median)
  // the 20-slice write-back's tail (App. A Q2/Q3): cells n-201 .. n-201+r-1 carry the rescaled depth of the last r bases,
  // computed with the fresh edge window [n-201, n-1]; the last r bases keep their raw depth
  const int64_t S20 = n / 20, r20 = n - 20 * S20;
  const int64_t zone = n - 201;             // no fast-path tile may reach this base (also covers the clamped windows, i >= n-101)

  auto rescale = [&](int d, uint32_t g) { return (int)((double)d * rdmean / s_table[g] + 0.5); };   // gccontent.cpp:89, truncation
  auto slow_value = [&](int64_t i) -> int {   // the value K3 + its tail fixup would have left at source index i
    if (r20 >= 2 && i >= n - 201 && i < n - 201 + r20) return rescale(depth[20 * S20 + (i - (n - 201))], (uint32_t)gc_count201(gcbits, n - 201));
    if (i >= 20 * S20) return depth[i];
    int64_t lo = i - 100;
    if (lo < 0) lo = 0;
    if (lo > n - 202) lo = n - 202;
    return rescale(depth[i], (uint32_t)gc_count201(gcbits, lo));
  };

  int k = 0;
  auto geometry = [&](int64_t tile, int64_t& P0, int64_t& P1, bool& fast, int64_t& soff) {
    P0 = tile * tile_elems;
    P1 = (P0 + tile_elems < ncompact) ? P0 + tile_elems : ncompact;
    while (k < nreg && R.brk(k) <= P0) ++k;   // tiles are visited in increasing order
    const bool plain = (k >= nreg) || (R.brk(k) >= P1);
    soff = P0 + R.shift(k);
    // fast: a whole tile, contiguous in the source, every base with an unclamped window, before the tail zone and before
    // the last partial stride of the 31 MAD residue classes
    fast = plain && P1 - P0 == tile_elems && P1 <= lim31 && soff >= 101 && soff + tile_elems <= zone;
  };
  // Chunks are aligned in the COMPACTED array (16 values = 64 aligned bytes of rdc, 16 aligned bytes of the LDS tile); in the
  // source they start at any byte, which a 16-byte load of the byte copy does not mind.
  uint4 regs[MAXC];
  uint64_t gwreg = 0;          // thread t < kK4GcWords: word t of the tile's staged GC words, committed to s_gw when the tile's turn comes
  int64_t gw0 = 0, gw0_next = 0;   // index of the first staged word of the tile being consumed / requested
  auto request = [&](int64_t soff) {   // branch-free: chunks beyond the tile re-read chunk 0 (ignored later)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int kc = c * kThreads + (int)threadIdx.x;
      const Bytes16 b = *reinterpret_cast<const Bytes16*>(d8 + soff + 16 * (kc < nchunks ? kc : 0));
      regs[c] = make_uint4(b.x, b.y, b.z, b.w);
    }
    gw0_next = (soff - 100) >> 6;   // soff >= 101 on this path
    const int64_t w = gw0_next + (int64_t)threadIdx.x;
    gwreg = gcbits[(threadIdx.x < kK4GcWords && w < nwords) ? w : 0];
  };

  int64_t P0, P1, soff; bool fast;
  int64_t tile = blockIdx.x;
  if (tile < ntiles) { geometry(tile, P0, P1, fast, soff); if (fast) request(soff); }
  for (; tile < ntiles; tile += gridDim.x) {
    // the staged GC words of this tile (requested with its data): nobody reads s_gw between the barrier in the middle of the
    // previous trip and the one below
    if (fast && threadIdx.x < kK4GcWords) { const int64_t w = gw0_next + (int64_t)threadIdx.x; s_gw[threadIdx.x] = w < nwords ? gwreg : 0; }
    gw0 = gw0_next;
    __syncthreads();   // s_val is free (and s_hist zeroed on the first trip); s_gw is complete
    if (fast) {
      const uint32_t p0mod = (uint32_t)(P0 % 31);
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const int kc = c * kThreads + (int)threadIdx.x;
        if (kc >= nchunks) continue;
        const uint32_t w4[4] = {regs[c].x, regs[c].y, regs[c].z, regs[c].w};
        // ---- the rescale K3' did: window GC count of the chunk's first base, one leaving / entering bit pair per base ----
        const int64_t p = soff + 16 * (int64_t)kc;                      // source index of the chunk's first base
        const uint32_t rel = (uint32_t)(p - 100 - (gw0 << 6));
        uint32_t cnt = gcw_window(s_gw, rel);
        const uint32_t leave = gcw_field16(s_gw, rel), enter = gcw_field16(s_gw, rel + 201);
        const bool esc = has_escape(w4[0]) || has_escape(w4[1]) || has_escape(w4[2]) || has_escape(w4[3]);
        const uint32_t cnt0 = cnt;
        bool redo = esc;
        int v[16];
        if (FIX) {
          unsigned int flags = 0;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const unsigned int R = s_rt[cnt];
            flags |= R;
            v[j] = (int)((__umul24((w4[j >> 2] >> (8 * (j & 3))) & 0xffu, R) + (1u << (kFixShift - 1))) >> kFixShift);   // the multiply looks at R's low 24 bits only
            cnt = cnt - ((leave >> j) & 1u) + ((enter >> j) & 1u);
          }
          redo = redo || (int)flags < 0;
        } else if (!esc) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            v[j] = (int)rescale_f32((float)((w4[j >> 2] >> (8 * (j & 3))) & 0xffu), s_ratio[cnt], redo);
            cnt = cnt - ((leave >> j) & 1u) + ((enter >> j) & 1u);
          }
        }
        if (redo) {   // rare: an escape byte (the lane's values come from the int32 array) or a value too close to a rounding
                      // boundary for the float form: the reference's own expression for the lane's sixteen bases
          cnt = cnt0;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int x = esc ? depth[p + j] : (int)((w4[j >> 2] >> (8 * (j & 3))) & 0xffu);
            v[j] = rescale(x, cnt);
            v[j] = v[j] < 0 ? 0 : v[j];   // an escaped depth is any int32 (negative ones are refused by the caller, K2j's flag)
            cnt = cnt - ((leave >> j) & 1u) + ((enter >> j) & 1u);
          }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = v[j] > capval ? capval : v[j];   // (never negative: depth bytes and ratios are not)
        uint32_t pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pk[q] = (uint32_t)v[4 * q] | ((uint32_t)v[4 * q + 1] << 8) | ((uint32_t)v[4 * q + 2] << 16) | ((uint32_t)v[4 * q + 3] << 24);
        *reinterpret_cast<uint4*>(s_val + 16 * kc) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4*>(rdc8 + P0 + 16 * (int64_t)kc) = make_uint4(pk[0], pk[1], pk[2], pk[3]);   // 16-byte aligned: P0 is a multiple of 64 m
        // sixteen LDS atomics into [value][MAD residue class]: the class of element j is cls0 + j, minus 31 from the lane's
        // wrap point on; the element index rides in the instruction's offset field
        const uint32_t cls0 = (p0mod + 16u * (uint32_t)kc) % 31u;
        unsigned int* ha = s_hist + cls0;
        if (cols == kK4jCols) {
#pragma unroll
          for (int j = 0; j < 16; ++j) atomicAdd(ha + v[j] * kK4jCols + j, 1u);   // column cls0 + j <= 45
        } else {   // 32 columns: minus 31 from the lane's wrap point on
          unsigned int* hb = ha - 31;
          const int jw = 31 - (int)cls0;
#pragma unroll
          for (int j = 0; j < 16; ++j) atomicAdd((j >= jw ? hb : ha) + v[j] * kResClasses + j, 1u);
        }
      }
    } else {
      // ---- per-element path: contiguous source segments between removed regions, values from the int32 array ----
      int kk = k;
      int64_t seg = P0;
      while (seg < P1) {
        const int64_t nxt = (kk < nreg && R.brk(kk) < P1) ? R.brk(kk) : P1;
        const int64_t len = nxt - seg;
        if (len > 0) {
          const int64_t so = seg + R.shift(kk);
          const int dst = (int)(seg - P0);
          for (int64_t e = threadIdx.x; e < len; e += kThreads) {
            int x = slow_value(so + e);
            if (x > capval) x = capval;
            if (x < 0) x = 0;   // negative depth is refused by the caller (K2's flag); keep the byte store in range
            s_val[dst + e] = (unsigned char)x;
            rdc8[seg + e] = (unsigned char)x;
            const int64_t p = seg + e;
            atomicAdd(&s_hist[x * cols + (p < lim31 ? (int)((uint32_t)p % 31u) : cols - 1)], 1u);
          }
        }
        seg = nxt;
        if (kk < nreg && R.brk(kk) == nxt) ++kk;
      }
    }
    __syncthreads();
    // ---- request the next tile now: its loads fly during the median phase ----
    const int64_t cur_tile = tile;
    if (tile + gridDim.x < ntiles) { geometry(tile + gridDim.x, P0, P1, fast, soff); if (fast) request(soff); }
    // ---- per-bin exact median (order statistic kth) and sum: `parts` threads per bin ----
    const int b_local = threadIdx.x / parts, part = threadIdx.x % parts;
    const int64_t b = cur_tile * TB + b_local;
    const bool active = b < nb;
    if (SW7) {
      // Values below 128 (the cap is): four to a register, straight from the LDS bytes.  A bin is the bytes [B, B + m) of the
      // tile; its (up to 27) dwords go round robin to the bin's four threads, bytes outside the bin masked -- to 0 for the
      // sum (v_sad_u8 adds four bytes in one instruction), to 0xff for the counts.  #{x > mid} of four values is one
      // subtraction and one popcount: with the top bit of every byte set, (x | 0x80) - (mid + 1) keeps that bit exactly
      // where x > mid, and no byte borrows from its neighbour.  Bisection from [0, cap]: the same seven steps for every bin.
      const int B = b_local * m;
      const int d0 = B >> 2, d1 = (B + m - 1) >> 2;
      const uint32_t* w = reinterpret_cast<const uint32_t*>(s_val);
      uint32_t xo[7];
      uint32_t ssum = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int d = d0 + part + parts * i;
        const uint32_t v = w[d <= d1 ? d : d1];
        const int lo_cut = B - 4 * d, hi_cut = 4 * d + 4 - (B + m);          // bytes of the dword before / after the bin
        uint32_t keep = 0xffffffffu;
        keep = lo_cut > 0 ? keep << (8 * lo_cut) : keep;
        keep = hi_cut > 0 ? keep & (0xffffffffu >> (8 * hi_cut)) : keep;
        keep = (d <= d1 && active) ? keep : 0u;
        ssum = __builtin_amdgcn_sad_u8(v & keep, 0u, ssum);
        xo[i] = (v | ~keep) | 0x80808080u;
      }
      ssum = (uint32_t)parts_sum((int)ssum, parts);
      // #{x <= t} of the bin, t = -1 .. 126 (masked bytes, 0xff, always count as "> t": 28 dword slots x 4 bytes - m of them per bin)
      auto count_le = [&](int t) {
        const uint32_t sub = (uint32_t)(t + 1) * 0x01010101u;
        int gt = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) gt += __popc((xo[i] - sub) & 0x80808080u);
        return 4 * 7 * parts - parts_sum(gt, parts);
      };
      // The median of a bin lies next to its mean: eight values around sum / m bracket it on all but a handful of bins (event
      // edges), and three bisection steps inside the bracket plus the two counts that prove it replace seven steps from
      // [0, cap].  A wave with a bin outside its bracket takes the seven steps (the result is the same either way).
      int lo = 0, hi = capval, steps = 7;
      {
        const int est = (int)((float)ssum * inv_m);
        int lo0 = est - 3;
        lo0 = lo0 < 0 ? 0 : lo0;
        int hi0 = lo0 + 7;
        hi0 = hi0 > capval ? capval : hi0;
        lo0 = lo0 > hi0 ? hi0 : lo0;
        const bool below = count_le(lo0 - 1) < kth;                                   // the median is not below the bracket
        const bool above = hi0 >= capval || count_le(hi0 > 126 ? 126 : hi0) >= kth;   // ... nor above it (every value is <= cap)
        if (__all((below && above) || !active)) { lo = lo0; hi = hi0; steps = 3; }
      }
#pragma unroll 1
      for (int it = 0; it < steps; ++it) {
        const int mid = (lo + hi) >> 1;
        const int le = count_le(mid);
        if (lo < hi) { if (le >= kth) hi = mid; else lo = mid + 1; }
      }
      if (active && part == 0) { binmed[b] = lo; binsum[b] = (int64_t)ssum; }
    } else if (EPT == 0) {
      // Caps of 128 .. 253 (a byte has no spare bit): the same scheme on 16-bit fields, two values to a register -- bytes 0
      // and 2 of a dword in one, bytes 1 and 3 in another, bit 15 of every field as the guard.  #{x > mid} of four values is
      // two subtractions and two popcounts; eight bisection steps from [0, cap].
      const int B = b_local * m;
      const int d0 = B >> 2, d1 = (B + m - 1) >> 2;
      const uint32_t* w = reinterpret_cast<const uint32_t*>(s_val);
      uint32_t xa[7], xc[7];
      uint32_t ssum = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int d = d0 + part + parts * i;
        const uint32_t v = w[d <= d1 ? d : d1];
        const int lo_cut = B - 4 * d, hi_cut = 4 * d + 4 - (B + m);          // bytes of the dword before / after the bin
        uint32_t keep = 0xffffffffu;
        keep = lo_cut > 0 ? keep << (8 * lo_cut) : keep;
        keep = hi_cut > 0 ? keep & (0xffffffffu >> (8 * hi_cut)) : keep;
        keep = (d <= d1 && active) ? keep : 0u;
        ssum = __builtin_amdgcn_sad_u8(v & keep, 0u, ssum);
        const uint32_t xb = v | ~keep;                                       // bytes outside the bin: 0xff, above every mid
        xa[i] = (xb & 0x00ff00ffu) | 0x80008000u;
        xc[i] = ((xb >> 8) & 0x00ff00ffu) | 0x80008000u;
      }
      ssum = (uint32_t)parts_sum((int)ssum, parts);
      auto count_le = [&](int t) {   // #{x <= t}, t = -1 .. 252; masked bytes (0xff) always count as "> t"
        const uint32_t sub = (uint32_t)(t + 1) * 0x00010001u;
        int gt = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) gt += __popc((xa[i] - sub) & 0x80008000u) + __popc((xc[i] - sub) & 0x80008000u);
        return 4 * 7 * parts - parts_sum(gt, parts);
      };
      // bracket of sixteen values around the bin's mean (deeper coverage: wider bins of values), four steps inside it
      int lo = 0, hi = capval, steps = 8;
      {
        const int est = (int)((float)ssum * inv_m);
        int lo0 = est - 7;
        lo0 = lo0 < 0 ? 0 : lo0;
        int hi0 = lo0 + 15;
        hi0 = hi0 > capval ? capval : hi0;
        lo0 = lo0 > hi0 ? hi0 : lo0;
        const bool below = count_le(lo0 - 1) < kth;
        const bool above = hi0 >= capval || count_le(hi0 > 252 ? 252 : hi0) >= kth;
        if (__all((below && above) || !active)) { lo = lo0; hi = hi0; steps = 4; }
      }
#pragma unroll 1
      for (int it = 0; it < steps; ++it) {
        const int mid = (lo + hi) >> 1;
        const int le = count_le(mid);
        if (lo < hi) { if (le >= kth) hi = mid; else lo = mid + 1; }
      }
      if (active && part == 0) { binmed[b] = lo; binsum[b] = (int64_t)ssum; }
    }
  }
  __syncthreads();
  // ---- per-workgroup histogram slab; the last workgroup folds them into res_hist (every value is below vr: overwrite)
  // and hands [BinAccum | histogram] to the host ----
  unsigned int* slab = hist_slabs + (size_t)blockIdx.x * vr * kResClasses;
  for (int e = threadIdx.x; e < vr * kResClasses; e += kThreads) {
    const unsigned int* row = s_hist + (e >> 5) * cols;
    const int c = e & 31;
    st_cg(&slab[e], cols == kResClasses ? row[c] : (c == 31 ? row[kK4jCols - 1] : row[c] + (c < 16 ? row[31 + c] : 0u)));
  }
  // res_hist is zero when the launch begins (K1's FillList): the groups' sums go straight into it
  if (!fold_slabs_add(hist_slabs, res_hist, vr * kResClasses, per_group, counters)) return;
  export_words(exp_dst, exp_src, exp_bytes);
}

inline int grid_for(int64_t items, int per_block) {
  int64_t g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > kMaxGrid) g = kMaxGrid;
  return (int)g;
}

}  // namespace

void report_attribute_failure(const char* kernel, const char* what) {
  fprintf(stderr, "librsi_hot: %s failed for %s: its launches with more than 48 KB of dynamic LDS will be rejected\n", what, kernel);
}
void launch_fasta_classify(const uint8_t* fasta, int64_t n, uint64_t* gcbits, uint64_t* nbits, int64_t nwords,
                           const FillList& fill, hipStream_t stream) {
  RSI_LAUNCH(k_fasta_classify, dim3(grid_for(nwords * 4, kThreads)), dim3(kThreads), 0, stream, fasta, n, gcbits,
                     nbits, nwords, fill);
}
__global__ __launch_bounds__(kThreads) void k_fill(FillList fill) { fill_ranges(fill); }
void launch_fill(const FillList& fill, hipStream_t stream) {
  unsigned long long units = 0;
  for (int k = 0; k < fill.n; ++k) units += fill.units[k];
  if (units == 0) return;
  RSI_LAUNCH(k_fill, dim3(grid_for((int64_t)units, kThreads)), dim3(kThreads), 0, stream, fill);
}
void launch_n_transitions(const uint64_t* nbits, int64_t nwords, uint64_t* list, uint32_t* count, uint32_t cap, int64_t n, int dx,
                          PhaseParams* pp, int64_t* cbreak, int64_t* cum, unsigned int* counter, hipStream_t stream) {
  // 256 workgroups: the launch ends with one same-address arrival atomic per workgroup (12 ns each: 2048 of them were 25 us of
  // a kernel that reads 16 MB)
  RSI_LAUNCH(k_n_transitions, dim3(std::min(grid_for(nwords, kThreads), 256)), dim3(kThreads), 0, stream, nbits, nwords, list,
                     count, cap, n, dx, pp, cbreak, cum, counter);
}
static int gc_hist_grid(int64_t n) {
  const int64_t nsub = (n + kSubBases - 1) / kSubBases;
  int64_t grid = (nsub + 3) / 4;
  if (grid > kMaxGrid) grid = kMaxGrid;
  const int64_t need = (nsub + kGcMaxSubPerWg - 1) / kGcMaxSubPerWg;
  if (grid < need) grid = need;
  return (int)(grid < 1 ? 1 : grid);
}
size_t gc_hist_slab_bytes(int64_t n) { return (size_t)gc_hist_grid(n) * kGcSlab * 8; }
size_t fold_scratch_bytes() { return (size_t)kFoldGroups * kK4Window * kResClasses * 4; }   // the widest slab (K4 at vr = 512)
void launch_gc_hist(const int32_t* depth, const uint64_t* gcbits, int64_t n, GcAccum* acc, double* table, int packed, void* slabs,
                    void* gsum, unsigned int* counters, uint8_t* depth8, hipStream_t stream) {
  const int grid = gc_hist_grid(n);
  unsigned long long* sl = static_cast<unsigned long long*>(slabs);
  unsigned long long* gs = static_cast<unsigned long long*>(gsum);
  const int pg = fold_per_group(grid);
  if (packed) RSI_LAUNCH(k_gc_hist<true>, dim3((unsigned)grid), dim3(kThreads), 0, stream, depth, gcbits, n, n / 64 + 1, sl, gs, pg, counters, acc, table, depth8);
  else RSI_LAUNCH(k_gc_hist<false>, dim3((unsigned)grid), dim3(kThreads), 0, stream, depth, gcbits, n, n / 64 + 1, sl, gs, pg, counters, acc, table, depth8);
}
static int gc_joint_grid(int64_t n) {
  const int64_t nsub = (n + kSubBases - 1) / kSubBases;
  int64_t grid = (nsub + kJWaves - 1) / kJWaves;
  if (grid > 256) grid = 256;   // one workgroup per CU: its histogram takes 107 KB of LDS
  return (int)(grid < 1 ? 1 : grid);
}
size_t gc_joint_slab_bytes(int64_t n) { return (size_t)gc_joint_grid(n) * kJSlabStride * 4; }
size_t gc_joint_totals_bytes() { return (size_t)kJTotWords * 4; }
size_t gc_joint_esc_list_bytes() { return (size_t)256 * (1 + kJEscPerWg) * 4; }
void launch_gc_joint_hist(const int32_t* depth, const uint64_t* gcbits, int64_t n, GcAccum* acc, double* table, void* slabs, void* totals,
                          unsigned int* counters, uint8_t* depth8, uint32_t* hist, ValueHistAux* aux, ValueMedian* vm,
                          const void* head_src, void* head_dst, size_t head_bytes, void* esc_list, unsigned int* rtab, JointInfo* info,
                          PhaseParams* pp, double cap_mult, hipStream_t stream, const NRuns* nruns) {
  const int grid = gc_joint_grid(n);
  NRuns nr{};
  if (nruns) nr = *nruns;
  RSI_ALLOW_FULL_LDS(k_gc_joint_hist);
  RSI_LAUNCH(k_gc_joint_hist, dim3((unsigned)grid), dim3(kJThreads), 0, stream, depth, gcbits, n, n / 64 + 1, static_cast<unsigned int*>(slabs),
             static_cast<unsigned int*>(totals), fold_per_group(grid), counters, acc, table, depth8, hist, aux, vm, head_src, head_dst,
             (unsigned int)head_bytes, getenv("RSI_HOT_K2J_DBG") ? atoi(getenv("RSI_HOT_K2J_DBG")) : 0, static_cast<unsigned int*>(esc_list), rtab, info,
             byte_escape_limit(n), pp, cap_mult, nr);
}
void launch_escape_hist(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table, uint32_t* hist,
                        ValueHistAux* aux, unsigned int* counter, ValueMedian* vm, const void* head_src, void* head_dst, size_t head_bytes,
                        hipStream_t stream) {
  RSI_LAUNCH(k_escape_hist, dim3(grid_for((n + 15) / 16, kThreads)), dim3(kThreads), 0, stream, depth8, depth, gcbits, n, table, hist, aux, counter,
             vm, head_src, head_dst, (unsigned int)head_bytes);
}
size_t gc_rescale_slab_bytes(int64_t n) { return (size_t)grid_for(n, kTileBases) * kDeepWidth * 4; }   // the wider of K3's two windows

__global__ void k_gc_tail_fixup_out(const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits, int64_t n,
                                    const double* __restrict__ table, int32_t* __restrict__ out);
void launch_gc_rescale(const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                       int adjust, int32_t* out, uint32_t* hist, ValueHistAux* aux, void* slabs, void* gsum,
                       unsigned int* counters, ValueMedian* vm, const void* head_src, void* head_dst, size_t head_bytes,
                       hipStream_t stream, uint8_t* out8, PhaseParams* pp, double cap_mult) {
  const int grid = grid_for(n, kTileBases);
  const dim3 g(grid), b(kThreads);
  unsigned int* sl = static_cast<unsigned int*>(slabs);
  unsigned int* gs = static_cast<unsigned int*>(gsum);
  const int pg = fold_per_group(grid);
  // adjust = 1 (rescaled array + histogram in one pass) is the deep-coverage path; adjust = 0, out = NULL the -NOGC histogram.
  // The tail cells of out[] get their quirks from a launch of their own (see k_gc_tail_fixup_out).
  if (adjust) {
    RSI_LAUNCH(k_gc_rescale<true>, g, b, 0, stream, depth, gcbits, n, n / 64 + 1, table, out, hist, aux, sl, gs, pg, counters, vm, head_src, head_dst, (unsigned int)head_bytes, 0, static_cast<uint8_t*>(nullptr), pp, cap_mult);
    if (out) RSI_LAUNCH(k_gc_tail_fixup_out, dim3(1), dim3(64), 0, stream, depth, gcbits, n, table, out);
  } else RSI_LAUNCH(k_gc_rescale<false>, g, b, 0, stream, depth, gcbits, n, n / 64 + 1, table, out, hist, aux, sl, gs, pg, counters, vm, head_src, head_dst, (unsigned int)head_bytes, 0, out8, pp, cap_mult);
}
// The tail quirks in out[] as a launch of its own: the cells it rewrites were written by other workgroups of the streaming
// launch, and two stores to one address from different XCDs within one launch have no defined order.
__global__ void k_gc_tail_fixup_out(const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits, int64_t n,
                                    const double* __restrict__ table, int32_t* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x < 64) gc_tail_fixup(depth, gcbits, n, table, 1, out, nullptr, nullptr);
}
void launch_gc_materialize(const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table, int32_t* out,
                           unsigned int* counter, hipStream_t stream) {
  const int grid = grid_for(n, kTileBases);
  RSI_LAUNCH(k_gc_rescale<true>, dim3(grid), dim3(kThreads), 0, stream, depth, gcbits, n, n / 64 + 1, table, out, nullptr, nullptr, nullptr,
                     nullptr, 1, counter, nullptr, nullptr, nullptr, 0u, 1, static_cast<uint8_t*>(nullptr), static_cast<PhaseParams*>(nullptr), 0.0);
  RSI_LAUNCH(k_gc_tail_fixup_out, dim3(1), dim3(64), 0, stream, depth, gcbits, n, table, out);
}
static int value_hist8_grid(int64_t n) {
  const int64_t nsub = (n + kSubBases - 1) / kSubBases;
  int64_t grid = (nsub + 3) / 4;
  if (grid > 256 * 4) grid = 256 * 4;
  return (int)(grid < 1 ? 1 : grid);
}
size_t value_hist8_slab_bytes(int64_t n) { return (size_t)value_hist8_grid(n) * kK3Width * 4; }
void launch_value_hist8(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                        uint32_t* hist, ValueHistAux* aux, void* slabs, void* gsum, unsigned int* counters, ValueMedian* vm,
                        const void* head_src, void* head_dst, size_t head_bytes, uint8_t* rescaled8, const unsigned int* escapes,
                        hipStream_t stream) {
  const int grid = value_hist8_grid(n);
  RSI_LAUNCH(k_value_hist8, dim3(grid), dim3(kThreads), 0, stream, depth8, depth, gcbits, n, n / 64 + 1, table, hist, aux,
                     static_cast<unsigned int*>(slabs), static_cast<unsigned int*>(gsum), fold_per_group(grid), counters, vm, head_src, head_dst,
                     (unsigned int)head_bytes, rescaled8, escapes, byte_escape_limit(n));
}

static void k4_geometry(int m, int32_t capval, int64_t ncompact, int vbase, int& TB, int& vr, int& grid) {
  TB = 64;   // bins per tile: as many as fit ~48 KB of values, 4..64, power of two
  while (TB > 4 && (size_t)TB * m * 4 > 48 * 1024) TB >>= 1;
  vr = 64;   // LDS histogram range: covers the capped values when the cap is active, 64..256
  while (vr < 256 && vr <= capval) vr <<= 1;
  if (vbase > 0) vr = kK4Window;   // deep coverage: a window of 512 values around the median (64 KB of LDS, one workgroup per CU)
  const int64_t ntiles = (ncompact + (int64_t)TB * m - 1) / ((int64_t)TB * m);
  grid = (int)(ntiles < 256 * 3 ? (ntiles < 1 ? 1 : ntiles) : 256 * 3);
}
size_t cap_compact_slab_bytes(int m, int32_t capval, int64_t ncompact, int vbase) {
  int TB, vr, grid;
  k4_geometry(m, capval, ncompact, vbase, TB, vr, grid);
  return (size_t)grid * vr * kResClasses * 4;
}
// K4' applies when the cap keeps every value in a byte below the escape code and the bin fits the register median phase.
int cap_compact8_applies(int m, int32_t capval) { return capval >= 1 && capval < kByteSat && m <= 440 ? 1 : 0; }
// Bins per tile: 64 with four threads per bin; 128 with two threads per bin for small bins (m <= 52: a bin is at most 14 dwords,
// seven per thread), so that a tile still holds ~6500 values -- at -m 51 tiles of 64 bins were half as long, twice as many
// barriers and median phases per base.  Wide bins the other way: 32 bins with eight threads per bin up to m = 216 (a bin spans at
// most (m + 3) / 4 + 1 = 55 dwords, seven per thread), 16 bins with sixteen threads up to m = 440 -- round 2 sent everything
// above m = 104 through the int32 kernels (-m 201: 299 us of K3 + K4 per 60 Mb against 109).
static int k48_bins_per_tile(int m) { return m <= 52 ? 128 : (m <= 104 ? 64 : (m <= 216 ? 32 : 16)); }
static void k48_geometry(int m, int32_t capval, int64_t ncompact, int& vr, int& grid, int& maxc) {
  vr = 64;
  while (vr < 256 && vr <= capval) vr <<= 1;
  const int64_t tile = (int64_t)k48_bins_per_tile(m) * m;
  const int64_t ntiles = (ncompact + tile - 1) / tile;
  grid = (int)(ntiles < 256 * 4 ? (ntiles < 1 ? 1 : ntiles) : 256 * 4);   // 3, 5 or 6 per CU: 2-5 % slower
  maxc = tile / 16 <= kThreads ? 1 : 2;
}
size_t cap_compact8_slab_bytes(int m, int32_t capval, int64_t ncompact) {
  int vr, grid, maxc;
  k48_geometry(m, capval, ncompact, vr, grid, maxc);
  return (size_t)grid * vr * kResClasses * 4;
}
void launch_cap_compact_bin8(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                             const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg, int64_t ncompact, int32_t capval,
                             int m, uint8_t* rdc, int32_t* binmed, int64_t* binsum, uint32_t* res_hist, void* slabs, void* gsum,
                             unsigned int* counters, const void* exp_src, void* exp_dst, size_t exp_bytes, hipStream_t stream, int raw) {
  int vr, grid, maxc;
  k48_geometry(m, capval, ncompact, vr, grid, maxc);
  const int TB = k48_bins_per_tile(m);
  const size_t lds = (size_t)maxc * kThreads * 16 + (size_t)vr * kResClasses * 4;
  unsigned int* sl = static_cast<unsigned int*>(slabs);
  unsigned int* gs = static_cast<unsigned int*>(gsum);
  const int pg = fold_per_group_add(grid);
  const int ept = (m + 3) / 4;
#define RSI_K48(MC, EP, SW) do { RSI_ALLOW_FULL_LDS((k_cap_compact_bin8<MC, EP, SW>));                                                  \
    RSI_LAUNCH((k_cap_compact_bin8<MC, EP, SW>), dim3(grid), dim3(kThreads), lds, stream, depth8, depth, gcbits, n, n / 64 + 1, table, \
                       cbreak, cum, nreg, ncompact, capval, m, TB, vr, rdc, binmed, binsum, res_hist, sl, gs, pg, counters,            \
                       exp_src, exp_dst, (unsigned int)exp_bytes, inl, raw); } while (0)
  const bool sw7 = capval <= 127;   // four values to a register in the median phase (k_cap_compact_bin8, SW7)
  (void)ept;
  if (sw7) { if (maxc == 1) RSI_K48(1, 1, true); else RSI_K48(2, 1, true); }
  else { if (maxc == 1) RSI_K48(1, 0, false); else RSI_K48(2, 0, false); }   // caps of 128 .. 253: two values to a register (EPT = 0)
#undef RSI_K48
}
// K4j: the same from K2j's byte copy of the RAW depth, rescaling on the way (no K3' in front)
void launch_rescale_compact_bin8(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                                 const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg, int64_t ncompact, int32_t capval,
                                 int m, uint8_t* rdc, int32_t* binmed, int64_t* binsum, uint32_t* res_hist, void* slabs, void* gsum,
                                 unsigned int* counters, const void* exp_src, void* exp_dst, size_t exp_bytes, const unsigned int* rtab,
                                 PhaseParams* pp, hipStream_t stream) {
  int vr, grid, maxc;
  k48_geometry(m, capval, ncompact, vr, grid, maxc);   // pp != NULL: capval is the caller's guess (it fixes vr and SW7), ncompact an upper bound
  const int TB = k48_bins_per_tile(m);
  const size_t lds = (size_t)maxc * kThreads * 16 + (size_t)vr * (vr <= 128 ? kK4jCols : kResClasses) * 4;
  unsigned int* sl = static_cast<unsigned int*>(slabs);
  unsigned int* gs = static_cast<unsigned int*>(gsum);
  const int pg = fold_per_group_add(grid);
#define RSI_K48J(MC, EP, SW, FX) do { RSI_ALLOW_FULL_LDS((k_rescale_compact_bin8<MC, EP, SW, FX>));                                      \
    RSI_LAUNCH((k_rescale_compact_bin8<MC, EP, SW, FX>), dim3(grid), dim3(kThreads), lds, stream, depth8, depth, gcbits, n, n / 64 + 1, table, \
               cbreak, cum, nreg, ncompact, capval, m, TB, vr, rdc, binmed, binsum, res_hist, sl, gs, pg, counters,            \
               exp_src, exp_dst, (unsigned int)exp_bytes, inl, rtab, pp); } while (0)
  const bool sw7 = capval <= 127;   // four values to a register in the median phase (SW7)
  const bool fix = rtab != nullptr;   // the caller hands the ratios over only when K2j verified them (pipeline.hip)
  if (fix) {
    if (sw7) { if (maxc == 1) RSI_K48J(1, 1, true, true); else RSI_K48J(2, 1, true, true); }
    else { if (maxc == 1) RSI_K48J(1, 0, false, true); else RSI_K48J(2, 0, false, true); }
  } else {
    if (sw7) { if (maxc == 1) RSI_K48J(1, 1, true, false); else RSI_K48J(2, 1, true, false); }
    else { if (maxc == 1) RSI_K48J(1, 0, false, false); else RSI_K48J(2, 0, false, false); }
  }
#undef RSI_K48J
}
// K4w applies under a cap of 254 .. 32766 with bins the register median phase holds, while no 16-bit counter of a workgroup's
// window can wrap (a workgroup's share, one class of one value: share / 31 + one per tile)
static void k416_geometry(int m, int64_t ncompact, int& grid, int& maxc) {
  const int64_t tile = (int64_t)k48_bins_per_tile(m) * m;
  const int64_t ntiles = (ncompact + tile - 1) / tile;
  grid = (int)(ntiles < 256 * 3 ? (ntiles < 1 ? 1 : ntiles) : 256 * 3);
  maxc = tile / 16 <= kThreads ? 1 : 2;
}
int cap_compact16_applies(int m, int32_t capval, int64_t ncompact) {
  if (!(capval >= kByteSat && capval < 32767 && m <= 440 && (int64_t)m * capval < (int64_t)1 << 31)) return 0;
  int grid, maxc;
  k416_geometry(m, ncompact, grid, maxc);
  const int64_t tile = (int64_t)k48_bins_per_tile(m) * m;
  const int64_t tiles_per_wg = ((ncompact + tile - 1) / tile + grid - 1) / grid;
  return tiles_per_wg * (tile / 31 + 2) < 60000 ? 1 : 0;
}
size_t cap_compact16_slab_bytes(int m, int64_t ncompact) {
  int grid, maxc;
  k416_geometry(m, ncompact, grid, maxc);
  return (size_t)grid * kK4Window * kResClasses * 4;
}
void launch_cap_compact_bin16(const int32_t* src, const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg, int64_t ncompact,
                              int32_t capval, int m, int32_t* rdc, int32_t* binmed, int64_t* binsum, uint32_t* res_hist, void* slabs,
                              unsigned int* counters, const void* exp_src, void* exp_dst, size_t exp_bytes, int vbase, hipStream_t stream) {
  int grid, maxc;
  k416_geometry(m, ncompact, grid, maxc);
  const int TB = k48_bins_per_tile(m);
  const size_t lds = (size_t)maxc * kThreads * 32 + (size_t)kK4Window * kResClasses * 2;
  const int pg = fold_per_group_add(grid);
#define RSI_K416(MC) do { RSI_ALLOW_FULL_LDS((k_cap_compact_bin16<MC>));                                                         \
    RSI_LAUNCH((k_cap_compact_bin16<MC>), dim3(grid), dim3(kThreads), lds, stream, src, cbreak, cum, nreg, ncompact, capval, m, TB, vbase, \
               rdc, binmed, binsum, res_hist, static_cast<unsigned int*>(slabs), pg, counters, exp_src, exp_dst, (unsigned int)exp_bytes, inl); } while (0)
  if (maxc == 1) RSI_K416(1); else RSI_K416(2);
#undef RSI_K416
}
unsigned int byte_escape_limit(int64_t n) { return (unsigned int)(n >> 3 > 0xffffffffll ? 0xffffffffll : n >> 3); }
int hist_window_base(double center, int width) {   // [base, base + width) around the centre of the distribution; 0 up to ~160x
  if (!(center >= 160.0)) return 0;
  const double b = center - (double)(3 * width / 8);
  return b > (double)(kHistValues - width) ? kHistValues - width : (b < 0.0 ? 0 : (int)b);
}
int cap_compact_overwrites(int m, int32_t capval, int64_t ncompact, int vbase) {
  int TB, vr, grid;
  k4_geometry(m, capval, ncompact, vbase, TB, vr, grid);
  return vbase == 0 && capval < vr ? 1 : 0;
}
void launch_cap_compact_bin(const int32_t* src, int64_t n, const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg,
                            int64_t ncompact, int32_t capval, int m, int32_t* rdc, int32_t* binmed, int64_t* binsum,
                            uint32_t* res_hist, BinAccum* acc, void* slabs, void* gsum, unsigned int* counters,
                            const void* exp_src, void* exp_dst, size_t exp_bytes, int vbase, hipStream_t stream) {
  int TB, vr, grid;
  k4_geometry(m, capval, ncompact, vbase, TB, vr, grid);
  const size_t tile_pad = ((size_t)TB * m + 3) & ~(size_t)3;
  // deep coverage (a window that follows the depth): 16-bit LDS counters while a workgroup's share cannot make one wrap
  const int pack16 = vbase > 0 && (ncompact + grid - 1) / grid < (int64_t)60000 * 31 ? 1 : 0;   // (margin: a class gets at most one value more per tile)
  const size_t lds = tile_pad * 4 + (size_t)vr * kResClasses * (pack16 ? 2 : 4);
  const int quads = (int)(tile_pad / 4);
  const int maxv = (quads + kThreads - 1) / kThreads;          // 16-byte loads per thread and tile
  const int parts = kThreads / TB, ept = (m + parts - 1) / parts;   // values per thread in the median phase
  unsigned int* sl = static_cast<unsigned int*>(slabs);
  unsigned int* gs = static_cast<unsigned int*>(gsum);
  const int pg = fold_per_group(grid);
  const int overwrite = vbase == 0 && capval < vr ? 1 : 0;
#define RSI_K4(MV, EP) do { RSI_ALLOW_FULL_LDS((k_cap_compact_bin<MV, EP>));                                                            \
    RSI_LAUNCH((k_cap_compact_bin<MV, EP>), dim3(grid), dim3(kThreads), lds, stream, src, n, cbreak, cum, nreg,                \
                       ncompact, capval, m, TB, vr, vbase, rdc, binmed, binsum, res_hist, acc, sl, gs, pg, counters, overwrite,              \
                       exp_src, exp_dst, (unsigned int)exp_bytes, inl, pack16); } while (0)
  if (maxv <= 4 && ept <= 13) RSI_K4(4, 13);          // m <= 52 (e.g. -m 51)
  else if (maxv <= 8 && ept <= 26) RSI_K4(8, 26);     // m <= 104 (e.g. the default -m 101)
  else if (ept <= 52) RSI_K4(13, 52);                 // m <= 191 with 4 threads per bin, or fewer bins per tile
  else RSI_K4(13, 0);
#undef RSI_K4
}

}  // namespace rsik
