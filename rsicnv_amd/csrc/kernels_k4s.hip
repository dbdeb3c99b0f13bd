// kernels_k4s.hip -- K4 as two launches (round 5): K4s streams, K4m takes the bin medians.
//
// K4j (kernels_base.hip: k_rescale_compact_bin8) does the GC rescale, the cap, the compaction, the [value][MAD residue class]
// histogram AND the per-bin medians in one workgroup-wide tile loop: 113 registers, 38 KB of LDS, two barriers per tile, four
// waves per SIMD -- an instruction-issue kernel at 0.27 of the issue rate.  Here the same work is cut where its
// parts need different things:
//   * K4s `k_rescale_compact_stream` -- per base: window GC count (one leaving / entering bit pair), fixed-point rescale with
//     K2j's verified ratios, cap, 16-byte store of the compacted bytes, one LDS atomic into the residue-class histogram.
//     Wave-autonomous (sub-tiles of 1024 compacted positions, GC words in a per-wave LDS slot, no workgroup barrier in the loop),
//     no floating point in the loop, at most 64 registers: eight waves per SIMD.  The loop body has NO branch around a memory
//     operation: the compiler's wait-count pass then knows how many loads and stores are in flight and waits for the sub-tile
//     at hand only (with one conditional load in the loop it waited for everything, the prefetch of the next sub-tile included:
//     every trip paid a memory round trip).  A chunk of sixteen positions the loop cannot do in fixed point -- a sub-tile cut by a
//     removed region or at a chromosome end, an escape byte (depth >= 255), a window count within reach of a GC level whose ratio
//     did not verify -- is computed like any other, stored (garbage), counted with an increment of ZERO, and marked in LDS.
//     Behind the loop the workgroup's eight waves share the marked chunks: the reference's own double expression from the int32
//     depth (clamped windows, App. A Q1; the 20-slice tail quirks, Q2/Q3), byte store, LDS atomic.  Sub-tiles are dealt out so
//     that neighbours go to different workgroups: a soft-masked stretch marks a hundred sub-tiles in a row.
//     Replaces the rescale of gccontent.cpp:89, apply_cap loaddata.cpp:229-240, concatenate_data loaddata.cpp:48-85 and the MAD
//     subsamples' inputs rsi.cpp:1127-1143.
//   * K4m `k_bin_median8` -- per bin: exact median (order statistic (m + 1) / 2, rsi.cpp:1363-1379) and sum (rsi.cpp:1147-1153)
//     of the compacted bytes (L2 / Infinity-Cache warm).  No LDS, no barrier: `PARTS` lanes of a wave share a bin, seven dwords
//     each straight from the byte array, SWAR counting bisection as in K4j.
// Results are identical to K4j's (same arithmetic, same exact fall-backs); `RSI_HOT_K4SPLIT=0` selects K4j for A/B runs.  A
// chromosome without verified ratios (the three-pass chain behind wrapped pair counters) keeps K4j's float form.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include "kernels.h"
#include "device_util.h"
#include "per_base_device.h"

namespace rsik {

namespace {

constexpr int kS4Threads = 512;                 // eight waves; four workgroups per CU = 32 waves at <= 64 registers (1024 threads: 25 % slower)
constexpr int kS4Waves = kS4Threads / 64;
constexpr int kS4Sub = 2048;                    // compacted positions per wave trip: thirty-two per lane (sixteen: the per-trip work -- fields, scan, marks,
                                                // addresses -- was more instructions than the bases' own)
constexpr int kS4GcWords = 38;                  // staged mask words per wave: (63 + 2048 + 201 + 32) / 64 + 1 = 37, + 1 that stays zero
constexpr int kS4Cols = 64;                     // columns of the LDS histogram while the value range is <= 128 (see below)
constexpr int kS4Grid = 256 * 3, kS4GridLong = 256 * 4;
constexpr int kS4MaxTrips = 20;                 // sub-tiles per wave the marks in LDS have room for: 335 Mb with 8192 waves (longer chromosomes: K4j)
constexpr int kM4Threads = 256;
constexpr int kM4Grid = 256 * 8;
#ifndef K4S_ABL
#define K4S_ABL 0   // ablation switches for timing runs (results are wrong with any of them set)
#endif

struct S4Regs { uint4 b0, b1; uint64_t gw; };
// a value that is the same in every lane: make the compiler keep it in scalar registers
__device__ inline int64_t uniform_i64(int64_t x) {
  return (int64_t)((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)x) |
                   ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)x >> 32)) << 32));
}

// does the configuration a queued launch (pp != NULL) was given fit what the device found?  Uniform: every workgroup of both
// launches decides alike.
__device__ inline bool k4s_config_fits(const PhaseParams* pp, int vr, bool sw7, int m) {
  const int32_t capval = pp->capval;
  return pp->regions_ok && pp->nreg <= kRegLds && capval >= 1 && capval < kByteSat && capval < vr && sw7 == (capval <= 127) && pp->ncompact >= (int64_t)m * 8;
}

// position of the r-th (0-based) set bit of m, r < popcount(m)
__device__ inline int nth_set_bit(unsigned long long m, int r) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const int c = __popcll((m >> pos) & ((1ull << w) - 1));
    if (r >= c) { r -= c; pos += w; }
  }
  return pos;
}

// fold_slabs_add (device_util.h) for slabs of PACKED counters: word k of a row holds classes 2k (low half) and 2k + 1 (high half)
// -- a workgroup's count of one (value, class) cell stays far below 65536 (its share of the chromosome / 31), so the slab is half
// the bytes to write and to read back; the group's last workgroup unpacks while it sums (sixteen members can overflow a field) and
// adds the two sums to total[2k], total[2k + 1].  True in the workgroup that arrives last.
__device__ inline bool fold_slabs_add16(const unsigned int* slabs, unsigned int* total, int width /* packed words, multiple of 4 */, int per_group,
                                        unsigned int* counters) {
  __shared__ unsigned int s_flag16__;
  const int nblocks = (int)gridDim.x;
  const int g = (int)blockIdx.x / per_group;
  const int ngroups = (nblocks + per_group - 1) / per_group;
  const int members = (g + 1) * per_group <= nblocks ? per_group : nblocks - g * per_group;
  drain();
  __syncthreads();   // the slab's st_cg stores have completed
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(&counters[1 + g], 1u);
    const bool last = t == (unsigned int)members - 1u;
    if (last) atomicExch(&counters[1 + g], 0u);
    s_flag16__ = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_flag16__) return false;
  {
    const unsigned int* src = slabs + (size_t)g * per_group * width;
    const int last = __builtin_amdgcn_readfirstlane(members - 1);
    const __amdgpu_buffer_rsrc_t rs = coherent_buffer(src);
    const unsigned int sbytes = (unsigned int)width * 4u;
    constexpr int kFlight = 16;
    for (int q = threadIdx.x; q < width / 4; q += blockDim.x) {
      unsigned int lo4[4] = {0, 0, 0, 0}, hi4[4] = {0, 0, 0, 0};
      for (int k0 = 0; k0 <= last; k0 += kFlight) {
        u32x4 v[kFlight];
#pragma unroll
        for (int j = 0; j < kFlight; ++j) v[j] = ld_cg_buf_x4(rs, (unsigned int)q * 16u, (unsigned int)(k0 + j < last ? k0 + j : last) * sbytes);
#pragma unroll
        for (int j = 0; j < kFlight; ++j) if (k0 + j <= last) {
          const unsigned int w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) { lo4[c] += w[c] & 0xffffu; hi4[c] += w[c] >> 16; }
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (lo4[c]) atomicAdd(total + 2 * (4 * q + c), lo4[c]);
        if (hi4[c]) atomicAdd(total + 2 * (4 * q + c) + 1, hi4[c]);
      }
    }
  }
  drain();
  __syncthreads();   // this group's atomics have completed
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(&counters[0], 1u);
    const bool last = t == (unsigned int)ngroups - 1u;
    if (last) atomicExch(&counters[0], 0u);
    s_flag16__ = last ? 1u : 0u;
  }
  __syncthreads();
  return s_flag16__ != 0;
}

// `sw7_guess`: the median launch behind this one was shaped for a cap <= 127 (four values per register) -- part of what a
// queued launch must check, so that both decline together.  escapes: K2j's count of depths of 255 and more (GcAccum::escapes,
// device memory): zero on an ordinary chromosome, and the loop then does not look for escape bytes at all.
// COLS64: [vr][64] counters (vr <= 128) -- columns 0 .. 30 the MAD residue classes, 31 .. 45 classes 0 .. 14 AGAIN (a lane's
// sixteen consecutive classes then never wrap: no per-base select), 63 the class of the bases behind the last full stride of 31.
// With a row of 64 words the bank of a lane's counter is its class alone, whatever its value: the classes of 32 neighbouring
// lanes are distinct but for one pair, so a wave's atomic is all but conflict-free (K4j's 48 columns put odd values sixteen
// banks further).  Else (vr = 256) the plain 32 columns with a wrap select (64 KB otherwise).
struct S4Lds {
  unsigned int pad[16];                 // rt is addressed from up to 64 bytes below its start (see the walk)
  unsigned int rt[kGcLevels + 6];       // the levels' fixed-point ratios (K2j verified them: kernels_base.hip)
  unsigned int badbits[8];              // levels that occur and lack a verified ratio (bit 31 without bit 30)
  uint32_t gw[kS4Waves][2 * kS4GcWords + 2];   // per wave: the mask words under its sub-tile's windows, as dwords
  unsigned long long odd[kS4Waves][2 * kS4MaxTrips + 4];   // per wave, trip and kilobyte: the chunks (bit = lane) left to the exact pass behind the loop
  int64_t brk[kRegLds], cum[kRegLds + 1];
};
// RAW: the bytes are final values already (the raw depth under -NOGC, K3''s rescaled bytes on the three-pass chain; saturated at 254 =
// "this much or more", above any cap this kernel takes): no GC words, no windows, no ratios -- cap, compaction, histogram only.
template <bool COLS64, bool RAW>
__global__ __launch_bounds__(kS4Threads, 6) void k_rescale_compact_stream(
    const uint8_t* __restrict__ d8, const int32_t* __restrict__ depth, const uint64_t* __restrict__ gcbits, int64_t n, int64_t nwords,
    const double* __restrict__ table /* [kGcLevels] + rdmean */, const int64_t* __restrict__ cbreak, const int64_t* __restrict__ cum, int nreg,
    int64_t ncompact, int32_t capval, int m, int vr, int sw7_guess, uint8_t* __restrict__ rdc8, uint32_t* __restrict__ res_hist,
    unsigned int* __restrict__ hist_slabs, int per_group, unsigned int* __restrict__ counters, const void* exp_src, void* exp_dst,
    unsigned int exp_bytes, K4Regions inl, const unsigned int* __restrict__ rtab, const unsigned int* __restrict__ escapes,
    PhaseParams* __restrict__ pp) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned int* s_hist = reinterpret_cast<unsigned int*>(smem);
  constexpr int cols = COLS64 ? kS4Cols : kResClasses;
  __shared__ S4Lds L;
  if (pp) {
    if (!k4s_config_fits(pp, vr, sw7_guess != 0, m)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) pp->redo = 1;
      return;
    }
    nreg = pp->nreg; ncompact = pp->ncompact; capval = pp->capval;
  }
  for (int e = threadIdx.x; e < vr * cols; e += kS4Threads) s_hist[e] = 0;
  if (!RAW) for (int e = threadIdx.x; e < kGcLevels; e += kS4Threads) L.rt[e] = rtab[e];
  if (threadIdx.x < 8) L.badbits[threadIdx.x] = 0u;
  if (nreg <= kRegInline && !pp) {
    for (int e = threadIdx.x; e < nreg; e += kS4Threads) L.brk[e] = inl.brk[e];
    for (int e = threadIdx.x; e <= nreg; e += kS4Threads) L.cum[e] = inl.cum[e];
  } else {
    for (int e = threadIdx.x; e < kRegLds && e < nreg; e += kS4Threads) L.brk[e] = cbreak[e];
    for (int e = threadIdx.x; e <= kRegLds && e <= nreg; e += kS4Threads) L.cum[e] = cum[e];
  }
  __syncthreads();
  if (!RAW) for (int e = threadIdx.x; e < kGcLevels; e += kS4Threads) {
    const unsigned int r = L.rt[e];
    if ((r >> 31) & ~(r >> 30) & 1u) atomicOr(&L.badbits[e >> 5], 1u << (e & 31));
  }
  __syncthreads();
  // the removed regions: at most kRegLds of them, all in LDS (a select between an LDS and a global address is a FLAT load, whose
  // wait is for every memory operation in flight, the prefetch included); a chromosome with more takes K4j
  struct { const int64_t* b; const int64_t* c; __device__ int64_t brk(int k) const { return b[k]; } __device__ int64_t shift(int k) const { return c[k]; } } R{L.brk, L.cum};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* const sgw = L.gw[wave];
  const bool anybad = !RAW && (L.badbits[0] | L.badbits[1] | L.badbits[2] | L.badbits[3] | L.badbits[4] | L.badbits[5] | L.badbits[6]) != 0u;
  const bool anyesc = !RAW && *escapes != 0u;

  const int64_t lim31 = (ncompact / 31) * 31;
  const int64_t zone = n - 201;             // no fast sub-tile may reach this base (the tail quirks, the clamped windows i >= n-101)
  const int64_t nsub = (ncompact + kS4Sub - 1) / kS4Sub;
  const int64_t stride = (int64_t)gridDim.x * kS4Waves;
  const int64_t sub0 = (int64_t)wave * gridDim.x + blockIdx.x;   // neighbouring sub-tiles: different workgroups
  // fast: a whole sub-tile, contiguous in the source, every base with an unclamped window, before the tail zone and before the
  // last partial stride of the 31 MAD residue classes.  Any other sub-tile is loaded from a harmless address and marked whole.
  // Every wave makes the same, even number of trips; lane t works out trip t's source offset once, before the loop (a search of
  // the region table per trip cost two hundred scalar instructions): the loop reads its three words with v_readlane.
  const int ntrips = (int)((nsub + stride - 1) / stride);   // <= kS4MaxTrips (the launcher checks)
  int64_t g_soff = RAW ? 0 : 128;   // lane t: trip t  (n >= 4040 under GC adjustment, gccontent.cpp:66: the first kilobytes and their mask words exist; RAW: the byte array is padded by 2048)
  {
    const int64_t s = sub0 + (int64_t)lane * stride;
    if (lane < ntrips && s < nsub) {
      const int64_t P0 = s * kS4Sub, P1 = P0 + kS4Sub;
      int lo = 0, hi = nreg;   // regions with brk <= P0 (upper bound)
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (R.brk(mid) <= P0) lo = mid + 1; else hi = mid; }
      const bool plain = lo >= nreg || R.brk(lo) >= P1;
      const int64_t so = P0 + R.shift(lo);
      if (plain && P1 <= lim31 && (RAW ? so + kS4Sub <= n : (so >= 101 && so + kS4Sub <= zone))) g_soff = so | ((int64_t)1 << 62);   // bit 62: fast
    }
  }
  auto geometry = [&](int t, int64_t& soff, bool& fast) {   // t uniform; a trip behind the last: not fast
    const int tt = t < 63 ? t : 63;
    const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)g_soff, tt), hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)g_soff >> 32), tt);
    fast = (hi32 >> 30) & 1u;
    soff = (int64_t)(((uint64_t)(hi32 & 0x3fffffffu) << 32) | lo32);
  };
  // Sixteen compacted positions per lane = sixteen bytes at ANY byte address of the source copy (one load), sixteen aligned
  // bytes of the output (one store); the mask words under the sub-tile's windows by the first lanes.  Unconditional.
  // A trip = two neighbouring kilobytes of the compacted array, A and B: sixteen positions of each per lane, so that every load
  // and store instruction of a wave is one contiguous kilobyte (thirty-two CONSECUTIVE positions per lane made each instruction
  // touch every other 16 bytes of two kilobytes: the stores alone then cost what the rest of the trip did).
  auto request = [&](S4Regs& r, int64_t soff) {
    const int64_t so = (K4S_ABL & 8) ? (soff & ~(int64_t)15) : soff;
    const Bytes16 b0 = *reinterpret_cast<const Bytes16*>(d8 + so + 16 * (int64_t)lane);
    const Bytes16 b1 = *reinterpret_cast<const Bytes16*>(d8 + so + 1024 + 16 * (int64_t)lane);
    r.b0 = make_uint4(b0.x, b0.y, b0.z, b0.w);
    r.b1 = make_uint4(b1.x, b1.y, b1.z, b1.w);
    if (!RAW) {
      const int64_t w = ((soff - 100) >> 6) + lane;
      r.gw = gcbits[(lane < kS4GcWords && w < nwords) ? w : 0];
    }
  };
  const uint32_t lane16 = (16u * (uint32_t)lane) % 31u;
  // sixteen mask bits from staged bit r on: two dwords and a funnel shift
  auto field16 = [&](uint32_t r) { return __builtin_amdgcn_alignbit(sgw[(r >> 5) + 1], sgw[r >> 5], r & 31u) & 0xffffu; };

  // One sub-tile.  smod = P0 % 31 = (2 s) % 31 (2048 = 66 * 31 + 2), kept up incrementally: no 64-bit division per trip.
  auto trip = [&](const S4Regs& cur, int64_t s, int64_t soff, bool fast, uint32_t smod, int tripno) {
    if (K4S_ABL & 128) {   // ablation: a plain copy with the kernel's access pattern
      *reinterpret_cast<uint4*>(rdc8 + s * kS4Sub + 16 * (int64_t)lane) = cur.b0;
      *reinterpret_cast<uint4*>(rdc8 + s * kS4Sub + 1024 + 16 * (int64_t)lane) = make_uint4(cur.b1.x + (uint32_t)cur.gw, cur.b1.y, cur.b1.z, cur.b1.w);
      return;
    }
    uint32_t leaveA = 0, enterA = 0, leaveB = 0, enterB = 0, cntA = 0, cntB = 0;
    int gainA = 0, lossA = 0, gainB = 0, lossB = 0;
    if (!RAW) {
      const int64_t gw0 = (soff - 100) >> 6;
      const uint64_t myw = (gw0 + lane < nwords) ? cur.gw : 0ull;
      if (lane < kS4GcWords) { sgw[2 * lane] = (uint32_t)myw; sgw[2 * lane + 1] = (uint32_t)(myw >> 32); }
      __builtin_amdgcn_wave_barrier();
      const uint32_t r0 = (uint32_t)(soff - 100 - (gw0 << 6));          // 0 .. 63, uniform
      const uint32_t relA = r0 + 16u * (uint32_t)lane;                    // the lane's first window of A, in staged bits; B's: 1024 further
      leaveA = field16(relA); enterA = field16(relA + 201); leaveB = field16(relA + 1024); enterB = field16(relA + 1225);
      __builtin_amdgcn_wave_barrier();   // the slot is rewritten by the next trip
      // The lane's first window counts: the kilobyte's first plus what the lanes before it gained and lost -- ONE wave scan for both
      // kilobytes (net + 16 in 16-bit fields: six DPP adds) instead of five 64-bit popcounts with masks per lane and kilobyte.
      // A kilobyte's first window: its 201 bits lie in the words lanes 0 .. 4 (A) and 16 .. 20 (B) hold -- each counts its share
      // (shift and mask chosen by lane, no branch), five lane reads add them up.
      gainA = __popc(enterA); lossA = __popc(leaveA); gainB = __popc(enterB); lossB = __popc(leaveB);
      const int biased = (gainA - lossA + 16) | ((gainB - lossB + 16) << 16);
      const int excl = wave_incl_scan(biased) - biased;   // fields: the lanes' sums before this one, + 16 lane
      uint32_t cfirstA, cfirstB;
      {
        const uint32_t rem = 9u + r0;                                       // bits of the window behind the first three words: 9 .. 72
        const uint64_t m3 = rem >= 64u ? ~0ull : ((1ull << rem) - 1), m4 = rem > 64u ? ((1ull << (rem - 64u)) - 1) : 0ull;   // (scalar)
        const int role = lane & 15;
        const uint64_t mk = role == 3 ? m3 : (role == 4 ? m4 : ~0ull);
        const int share = __popcll((myw >> (role == 0 ? r0 : 0u)) & mk);
        cfirstA = (uint32_t)(__builtin_amdgcn_readlane(share, 0) + __builtin_amdgcn_readlane(share, 1) + __builtin_amdgcn_readlane(share, 2) +
                             __builtin_amdgcn_readlane(share, 3) + __builtin_amdgcn_readlane(share, 4));
        cfirstB = (uint32_t)(__builtin_amdgcn_readlane(share, 16) + __builtin_amdgcn_readlane(share, 17) + __builtin_amdgcn_readlane(share, 18) +
                             __builtin_amdgcn_readlane(share, 19) + __builtin_amdgcn_readlane(share, 20));
      }
      cntA = cfirstA + ((uint32_t)excl & 0xffffu) - 16u * (uint32_t)lane;
      cntB = cfirstB + ((uint32_t)excl >> 16) - 16u * (uint32_t)lane;
    }
    // An escape byte (the value is in the int32 array), or a window count within reach of a level whose ratio did not verify
    // (the thinly populated levels next to N runs and soft-masked stretches, whose mean depth is a mix: ratios of 4 and more):
    // the lane's sixteen positions of that kilobyte are left to the exact pass.  So is every lane of a sub-tile that is not `fast`.
    bool oddA = !fast, oddB = !fast;
    if (anyesc) {
      oddA = oddA || has_escape(cur.b0.x) || has_escape(cur.b0.y) || has_escape(cur.b0.z) || has_escape(cur.b0.w);
      oddB = oddB || has_escape(cur.b1.x) || has_escape(cur.b1.y) || has_escape(cur.b1.z) || has_escape(cur.b1.w);
    }
    if (anybad) {   // levels cnt - loss .. cnt + gain (at most 33) against the bitmap of such levels
      auto reach = [&](uint32_t cnt, int gain, int loss) {
        const uint32_t lo = (cnt - (uint32_t)loss) & 0xffu;   // (a sub-tile that is not fast counts garbage: keep the index inside the bitmap)
        const uint32_t lo5 = lo >> 5 > 6u ? 6u : lo >> 5;
        const uint64_t bits = ((uint64_t)L.badbits[lo5 + 1] << 32) | L.badbits[lo5];
        return ((bits >> (lo & 31u)) & ((1ull << (gain + loss + 1)) - 1)) != 0;
      };
      oddA = oddA || reach(cntA, gainA, lossA);
      oddB = oddB || reach(cntB, gainB, lossB);
    }
    const unsigned long long maskA = __ballot(oddA), maskB = __ballot(oddB);
    L.odd[wave][2 * tripno] = maskA; L.odd[wave][2 * tripno + 1] = maskB;   // (every lane, the same words: no branch)
    const unsigned int incA = oddA ? 0u : 1u, incB = oddB ? 0u : 1u;
    uint32_t clsA = smod + lane16;
    clsA = clsA >= 31u ? clsA - 31u : clsA;
    const uint32_t clsB = clsA == 30u ? 0u : clsA + 1u;   // 1024 = 33 * 31 + 1
    unsigned int* const haA = s_hist + clsA;
    unsigned int* const haB = s_hist + clsB;
    // The walk: the ratio of base j sits at rt[cnt_j], cnt_{j+1} = cnt_j + enter_j - leave_j.  With f_j = enter_j + (1 - leave_j) in
    // 2-bit fields (even bases in De, odd ones in Do, A in the low and B in the high half: no bit spreading needed) the pointer
    // rp_j = rt + cnt_j + j - 16 moves by f_j, and the read at rp_j[16 - j] has its constant in the instruction's offset field.
    // (A marked lane stays where it is.)
    const uint32_t nlA = ~leaveA & 0xffffu, nlB = ~leaveB & 0xffffu;
    const uint32_t De = (oddA ? 0u : (enterA & 0x5555u) + (nlA & 0x5555u)) | ((oddB ? 0u : (enterB & 0x5555u) + (nlB & 0x5555u)) << 16);
    const uint32_t Do = (oddA ? 0u : ((enterA >> 1) & 0x5555u) + ((nlA >> 1) & 0x5555u)) | ((oddB ? 0u : ((enterB >> 1) & 0x5555u) + ((nlB >> 1) & 0x5555u)) << 16);
    const unsigned int* rpA = L.rt + (oddA || RAW ? 100u : cntA) - 16;
    const unsigned int* rpB = L.rt + (oddB || RAW ? 100u : cntB) - 16;
    // A's four quads, then B's; the ratios of the next quad are requested before the current one is worked on: one LDS round
    // trip per trip in the open instead of eight
    unsigned int rq[4] = {0, 0, 0, 0}, rn[4] = {0, 0, 0, 0};
    auto fetch = [&](int qi, unsigned int (&r)[4]) {   // qi = 4 (B ? 1 : 0) + q
      const int q = qi & 3, hb = qi >> 2;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = 4 * q + t;
        const unsigned int*& rp = hb ? rpB : rpA;
        r[t] = (K4S_ABL & 2) ? 0x400000u + (unsigned)j : rp[16 - j];
        rp += __builtin_amdgcn_ubfe((j & 1) ? Do : De, 16 * hb + 2 * (j >> 1), 2);
      }
    };
    if (!RAW) fetch(0, rq);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      uint32_t pk[4];
      unsigned int* const ha = hb ? haB : haA;
      const unsigned int inc = hb ? incB : incA;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!RAW && 4 * hb + q < 7) fetch(4 * hb + q + 1, rn);
        const uint32_t w = hb ? (q == 0 ? cur.b1.x : q == 1 ? cur.b1.y : q == 2 ? cur.b1.z : cur.b1.w) : (q == 0 ? cur.b0.x : q == 1 ? cur.b0.y : q == 2 ? cur.b0.z : cur.b0.w);
        int v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // (byte * R + 2^21) >> 22, R < 2^24: the multiply looks at R's low 24 bits only
          const uint32_t x = RAW ? (w >> (8 * t)) & 0xffu : (__umul24((w >> (8 * t)) & 0xffu, rq[t]) + (1u << (kFixShift - 1))) >> kFixShift;
          v[t] = (int)(x > (uint32_t)capval ? (uint32_t)capval : x);
        }
        pk[q] = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        // LDS atomics into [value][MAD residue class]: the class of element j is cls + j (minus 31 from the lane's wrap point
        // on, folded back when the slab leaves); the element index rides in the instruction's offset field
        if (COLS64) {
#pragma unroll
          for (int t = 0; t < 4; ++t) if (!(K4S_ABL & 1)) atomicAdd(ha + v[t] * kS4Cols + (4 * q + t), inc);   // column cls + j <= 45
        } else {   // 32 columns: minus 31 from the lane's wrap point on
          const int jw = 31 - (int)(hb ? clsB : clsA);
#pragma unroll
          for (int t = 0; t < 4; ++t) atomicAdd((4 * q + t >= jw ? ha - 31 : ha) + v[t] * kResClasses + (4 * q + t), inc);
        }
        if (!RAW) {
#pragma unroll
          for (int t = 0; t < 4; ++t) rq[t] = rn[t];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the quads apart: hoisting all the byte extractions costs a register each
      }
      // (a marked lane's bytes are garbage until the exact pass rewrites them; the array is padded to whole sub-tiles)
      if (!(K4S_ABL & 16)) *reinterpret_cast<uint4*>(rdc8 + s * kS4Sub + 1024 * hb + 16 * (int64_t)lane) = make_uint4(pk[0], pk[1], pk[2], pk[3]);   // 16-byte aligned
    }
  };

  // Two register sets used alternately (a copy from one to the other would have to wait for the loads it copies), an even number
  // of trips for every wave: a trip behind the array's end is a marked one that nobody looks at.
  S4Regs ra, rb;
  int64_t soffa, soffb;
  bool fasta, fastb;
  uint32_t smod = (uint32_t)((2 * sub0) % 31);
  const uint32_t dmod = (uint32_t)((2 * stride) % 31);
  auto clampsub = [&](int64_t s) { return s < nsub ? s : nsub; };   // (a sub-tile behind the last: its bytes go to the array's padding)
  geometry(0, soffa, fasta);
  request(ra, soffa);
  for (int t = 0; t < ntrips; t += 2) {
    const int64_t s = sub0 + (int64_t)t * stride;
    geometry(t + 1, soffb, fastb);
    request(rb, soffb);
    trip(ra, clampsub(s), soffa, fasta, smod, t);
    smod += dmod; smod = smod >= 31u ? smod - 31u : smod;
    geometry(t + 2, soffa, fasta);
    request(ra, soffa);
    trip(rb, clampsub(s + stride), soffb, fastb, smod, t + 1);
    smod += dmod; smod = smod >= 31u ? smod - 31u : smod;
  }
  __syncthreads();
  // ---- the exact pass: the workgroup's marked chunks, per element, with the reference's own expression ----
  {
    const double rdmean = RAW ? 0.0 : table[kGcLevels];
    // the 20-slice write-back's tail (App. A Q2/Q3): cells n-201 .. n-201+r-1 carry the rescaled depth of the last r bases,
    // computed with the fresh edge window [n-201, n-1]; the last r bases keep their raw depth
    const int64_t S20 = n / 20, r20 = n - 20 * S20;
    auto rescale = [&](int d, uint32_t g) { return (int)((double)d * rdmean / table[g] + 0.5); };   // gccontent.cpp:89, truncation
    // One element in three steps, so that a thread can have the loads of several in flight: where it lies, its loads, its value.
    struct Elem { int64_t pc; int64_t lo; int d; uint64_t w[5]; bool ok, raw; };
    auto locate = [&](Elem& E, unsigned long long maskA, unsigned long long maskB, int nA, int total, int e, int64_t s) {
      E.ok = e < total;
      const bool inA = e < nA;
      const int ee = inA ? e : e - nA;
      const unsigned long long m = inA ? maskA : maskB;
      E.pc = s * kS4Sub + (inA ? 0 : 1024) + 16 * (int64_t)nth_set_bit(m, E.ok ? ee >> 4 : 0) + (ee & 15);
      E.ok = E.ok && E.pc < ncompact;
      const int64_t pc = E.ok ? E.pc : 0;
      int lo = 0, hi = nreg;   // regions with brk <= pc (upper bound)
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (R.brk(mid) <= pc) lo = mid + 1; else hi = mid; }
      const int64_t i = pc + R.shift(lo);   // source index; the value K3 + its tail fixup would have left there:
      const bool quirk = !RAW && r20 >= 2 && i >= n - 201 && i < n - 201 + r20;
      E.raw = RAW || (!quirk && i >= 20 * S20);
      E.d = depth[quirk ? 20 * S20 + (i - (n - 201)) : i];
      int64_t wl = i - 100;
      wl = wl < 0 ? 0 : wl;
      wl = wl > n - 202 ? n - 202 : wl;
      E.lo = quirk ? n - 201 : wl;
      // #GC in [lo, lo + 201): the five words the window can touch, their loads independent of each other
      const int64_t k = E.lo >> 6;
      const int64_t k4 = k + 4 < nwords ? k + 4 : nwords - 1;   // (touched only when the window reaches it, and then it exists)
      if (!RAW) { E.w[0] = gcbits[k]; E.w[1] = gcbits[k + 1]; E.w[2] = gcbits[k + 2]; E.w[3] = gcbits[k + 3]; E.w[4] = gcbits[k4]; }
      else { E.w[0] = E.w[1] = E.w[2] = E.w[3] = E.w[4] = 0; }
    };
    auto finish = [&](const Elem& E) {
      const uint32_t bsh = (uint32_t)(E.lo & 63);
      const uint32_t rem = 9 + bsh;                                   // bits of the window behind the first three words: 9 .. 72
      const uint64_t m3 = rem >= 64 ? ~0ull : ((1ull << rem) - 1);
      const uint64_t m4 = rem > 64 ? ((1ull << (rem - 64)) - 1) : 0ull;
      const uint32_t g = (uint32_t)(__popcll(E.w[0] >> bsh) + __popcll(E.w[1]) + __popcll(E.w[2]) + __popcll(E.w[3] & m3) + __popcll(E.w[4] & m4));
      int x = E.raw ? E.d : rescale(E.d, g);
      if (x > capval) x = capval;
      if (x < 0) x = 0;   // negative depth is refused by the caller (K2j's flag); keep the byte store in range
      if (E.ok) {
        rdc8[E.pc] = (unsigned char)x;
        atomicAdd(&s_hist[x * cols + (E.pc < lim31 ? (int)(E.pc % 31) : cols - 1)], 1u);
      }
    };
    for (int w = 0; w < kS4Waves; ++w) {
      for (int t = 0; t < ntrips; ++t) {
        const unsigned long long maskA = L.odd[w][2 * t], maskB = L.odd[w][2 * t + 1];   // uniform
        const int64_t s = (int64_t)w * gridDim.x + blockIdx.x + (int64_t)t * stride;
        if ((maskA | maskB) == 0ull || s >= nsub || (K4S_ABL & 64)) continue;
        const int nA = 16 * __popcll(maskA), total = nA + 16 * __popcll(maskB);
        for (int e0 = 0; e0 < total; e0 += 2 * kS4Threads) {   // two elements per thread and round, their loads in flight together
          Elem E0, E1;
          locate(E0, maskA, maskB, nA, total, e0 + (int)threadIdx.x, s);
          locate(E1, maskA, maskB, nA, total, e0 + kS4Threads + (int)threadIdx.x, s);
          finish(E0);
          finish(E1);
        }
      }
    }
  }
  __syncthreads();
  // ---- per-workgroup histogram slab ([vr][32], the repeated columns folded back); the last workgroup of every group adds the
  // group's sums to res_hist (zero when the launch begins: K1's FillList), the launch's last workgroup hands
  // [BinAccum | histogram] to the host ----
  unsigned int* slab = hist_slabs + (size_t)blockIdx.x * vr * (kResClasses / 2);
  auto cell = [&](int v, int c) -> unsigned int {
    const unsigned int* row = s_hist + v * cols;
    return !COLS64 ? row[c] : (c == 31 ? row[kS4Cols - 1] : row[c] + (c < 15 ? row[31 + c] : 0u));
  };
  for (int e = threadIdx.x; e < vr * (kResClasses / 2); e += kS4Threads) {
    const int v = e >> 4, c = 2 * (e & 15);
    st_cg(&slab[e], cell(v, c) | (cell(v, c + 1) << 16));   // (a workgroup's count per cell: below 65536, see fold_slabs_add16)
  }
  if (!fold_slabs_add16(hist_slabs, res_hist, vr * (kResClasses / 2), per_group, counters)) return;
  export_words(exp_dst, exp_src, exp_bytes);
}

// K4m: per bin the exact median (order statistic kth = (m + 1) / 2, m odd: rsi.cpp:2061, 1363-1379) and the sum
// (rsi.cpp:1147-1153) of the capped, compacted bytes.  A bin is the bytes [b m, b m + m) of rdc8; its (up to 7 * PARTS) dwords
// go round robin to the bin's PARTS lanes, bytes outside the bin masked -- to 0 for the sum (v_sad_u8 adds four bytes in one
// instruction), to 0xff for the counts.  SW7 (cap <= 127): #{x > t} of four values is one subtraction and one popcount -- with
// the top bit of every byte set, (x | 0x80) - (t + 1) keeps that bit exactly where x > t, and no byte borrows from its
// neighbour; caps of 128 .. 253: the same on 16-bit fields, two values to a register.  The median of a bin lies next to its
// mean: a bracket of eight (sixteen) values around sum / m holds it on all but a handful of bins (event edges); two counts
// prove the bracket, three (four) bisection steps finish inside it, a wave with a bin outside its bracket bisects [0, cap].
// The dwords of the NEXT trip are requested before the current trip's bisection: a wave always has seven loads in flight.
template <bool SW7, int PARTS>
__global__ __launch_bounds__(kM4Threads, SW7 ? 8 : 6) void k_bin_median8(
    const uint8_t* __restrict__ rdc8, int64_t ncompact, int32_t capval, int m, int vr_guess, int32_t* __restrict__ binmed,
    int64_t* __restrict__ binsum, const PhaseParams* __restrict__ pp) {
  if (pp) {   // queued behind K2j: the same test as the streaming launch's (which has raised pp->redo already when this fails)
    if (!k4s_config_fits(pp, vr_guess, SW7, m)) return;
    ncompact = pp->ncompact; capval = pp->capval;
  }
  constexpr int kBins = 64 / PARTS;   // bins per wave trip
  const int lane = threadIdx.x & 63;
  const int bl = lane / PARTS, part = lane % PARTS;
  const int64_t nb = ncompact / m;
  const int64_t ntrips = (nb + kBins - 1) / kBins;
  const int64_t stride = (int64_t)gridDim.x * (kM4Threads / 64);
  const int kth = (m + 1) / 2;
  const float inv_m = 1.0f / (float)m;      // for the first guess only (any guess gives the same median)
  const uint32_t* __restrict__ w32 = reinterpret_cast<const uint32_t*>(rdc8);
  // A trip's sixteen bins start at byte t kBins m = 4 q + r of the array (uniform: scalar arithmetic); everything a lane adds to
  // that is 32 bits wide.  x = r + bl m: the bin's first byte counted from dword q; its dwords are (x >> 2) + part + PARTS i,
  // unclamped (the array is padded; what lies outside the bin is masked).  Dword i of a lane lies wholly inside the bin whenever
  // i >= 1 and 4 (PARTS i + PARTS) <= m (five of the seven at m = 101): no mask to build for those.
  const int i_full = m / (4 * PARTS) - 1;   // the last such i
  auto geometry = [&](int64_t t, int64_t& q, uint32_t& x, bool& active) {
    const int64_t bin0 = t * kBins;
    const int64_t base = bin0 * m;
    q = base >> 2;
    int blc = bl;
    const int64_t left = nb - bin0;   // >= 1
    active = bl < left;
    blc = active ? bl : (int)left - 1;   // a lane without a bin repeats the last one (nothing stored)
    x = (uint32_t)(base & 3) + (uint32_t)blc * (uint32_t)m;
  };
  auto request = [&](int64_t t, uint32_t (&v)[7]) {
    int64_t q; uint32_t x; bool active;
    geometry(t, q, x, active);
    const uint32_t* p = w32 + q + (x >> 2) + part;
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] = p[PARTS * i];
  };
  auto process = [&](int64_t t, const uint32_t (&v)[7]) {
    int64_t q; uint32_t x; bool active;
    geometry(t, q, x, active);
    const int64_t b = t * kBins + bl;
    const int end = (int)(x + (uint32_t)m);   // first byte behind the bin, counted from dword q
    const int d0 = (int)(x >> 2) + part;
    uint32_t ssum = 0;
    uint32_t xa[7], xc[SW7 ? 1 : 7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      uint32_t xb = v[i];
      if (i >= 1 && i <= i_full) {   // uniform
        ssum = __builtin_amdgcn_sad_u8(xb, 0u, ssum);
      } else {
        const int d = d0 + PARTS * i;
        const int lo_cut = (int)x - 4 * d, hi_cut = 4 * d + 4 - end;          // bytes of the dword before / after the bin
        uint32_t keep = 0xffffffffu;
        keep = lo_cut > 0 ? keep << (8 * lo_cut) : keep;
        keep = hi_cut > 0 ? (hi_cut >= 4 ? 0u : keep & (0xffffffffu >> (8 * hi_cut))) : keep;
        ssum = __builtin_amdgcn_sad_u8(xb & keep, 0u, ssum);
        xb |= ~keep;                                                          // bytes outside the bin: 0xff, above every threshold
      }
      if (SW7) xa[i] = xb | 0x80808080u;
      else { xa[i] = (xb & 0x00ff00ffu) | 0x80008000u; xc[i] = ((xb >> 8) & 0x00ff00ffu) | 0x80008000u; }
    }
    ssum = (uint32_t)parts_sum((int)ssum, PARTS);
    // #{x <= t} of the bin (masked bytes always count as "> t": 28 byte slots per lane, minus the bin's m, over the bin's lanes)
    auto count_le = [&](int t) {
      int gt = 0;
      if (SW7) {
        const uint32_t sub = (uint32_t)(t + 1) * 0x01010101u;
#pragma unroll
        for (int i = 0; i < 7; ++i) gt += __popc((xa[i] - sub) & 0x80808080u);
      } else {
        const uint32_t sub = (uint32_t)(t + 1) * 0x00010001u;
#pragma unroll
        for (int i = 0; i < 7; ++i) gt += __popc((xa[i] - sub) & 0x80008000u) + __popc((xc[i] - sub) & 0x80008000u);
      }
      return 4 * 7 * PARTS - parts_sum(gt, PARTS);
    };
    constexpr int kHalf = SW7 ? 3 : 7, kTop = SW7 ? 126 : 252;
    int lo = 0, hi = capval, steps = SW7 ? 7 : 8;
    {
      const int est = (int)((float)ssum * inv_m);
      int lo0 = est - kHalf;
      lo0 = lo0 < 0 ? 0 : lo0;
      int hi0 = lo0 + 2 * kHalf + 1;
      hi0 = hi0 > capval ? capval : hi0;
      lo0 = lo0 > hi0 ? hi0 : lo0;
      const bool below = count_le(lo0 - 1) < kth;                                     // the median is not below the bracket
      const bool above = hi0 >= capval || count_le(hi0 > kTop ? kTop : hi0) >= kth;   // ... nor above it (every value is <= cap)
      if (__all((below && above) || !active)) { lo = lo0; hi = hi0; steps = SW7 ? 3 : 4; }
    }
#pragma unroll 1
    for (int it = 0; it < steps; ++it) {
      const int mid = (lo + hi) >> 1;
      const int le = count_le(mid);
      if (lo < hi) { if (le >= kth) hi = mid; else lo = mid + 1; }
    }
    if (active && part == 0) { binmed[b] = lo; binsum[b] = (int64_t)ssum; }
  };

  uint32_t va[7], vb[7];
  int64_t t = (int64_t)blockIdx.x * (kM4Threads / 64) + (threadIdx.x >> 6);
  // (requests are unconditional -- a trip behind the last asks for the last one again -- so that the compiler's wait counts stay exact)
  auto clampt = [&](int64_t u) { return u < ntrips ? u : ntrips - 1; };
  if (t < ntrips) request(t, va);
  while (t < ntrips) {
    request(clampt(t + stride), vb);
    process(t, va);
    t += stride;
    if (t >= ntrips) break;
    request(clampt(t + stride), va);
    process(t, vb);
    t += stride;
  }
}

static int k4s_vr(int32_t capval) {
  int vr = 64;
  while (vr < 256 && vr <= capval) vr <<= 1;
  return vr;
}
static int k4s_grid(int64_t ncompact) {
  const int64_t nsub = (ncompact + kS4Sub - 1) / kS4Sub;
  int64_t g = (nsub + kS4Waves - 1) / kS4Waves;
  // Three workgroups per CU are resident (80 registers: six waves per SIMD), so 768 workgroups are one round of the chip; a
  // fourth per CU would wait for a slot -- and so would every small kernel of the pool's other chromosomes, behind it (768
  // against 1024: the same time alone, 2 % on the pooled step).  Chromosomes too long for twenty trips per wave take 1024.
  const char* ge = getenv("RSI_HOT_K4S_GRID");   // (timing runs)
  const int gv = ge ? atoi(ge) : 0;
  int cap = gv >= 64 && gv <= 4096 ? gv : kS4Grid;
  if (!gv && (nsub + (int64_t)cap * kS4Waves - 1) / ((int64_t)cap * kS4Waves) > kS4MaxTrips) cap = kS4GridLong;
  g = g < 1 ? 1 : (g > cap ? cap : g);
  return (int)g;
}

}  // namespace

// 1 <= capval < kByteSat, bins the median phase holds, removed regions that fit the LDS table, and no more sub-tiles per wave than
// the marks in LDS have room for
int rescale_compact_split_applies(int m, int32_t capval, int64_t ncompact, int nreg) {
  if (!(capval >= 1 && capval < kByteSat && m <= 440 && nreg <= kRegLds)) return 0;
  const int64_t nsub = (ncompact + kS4Sub - 1) / kS4Sub;
  const int64_t stride = (int64_t)k4s_grid(ncompact) * kS4Waves;
  return (nsub + stride - 1) / stride <= kS4MaxTrips ? 1 : 0;
}
size_t rescale_compact_split_slab_bytes(int32_t capval, int64_t ncompact) {
  return (size_t)k4s_grid(ncompact) * k4s_vr(capval) * (kResClasses / 2) * 4;   // packed: two classes per word
}
size_t rescale_compact_split_rdc_bytes(int64_t ncompact) { return (size_t)(((ncompact + kS4Sub - 1) / kS4Sub + 1) * kS4Sub + 64); }

void launch_rescale_compact_stream(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                                   const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg, int64_t ncompact, int32_t capval,
                                   int m, uint8_t* rdc, uint32_t* res_hist, void* slabs, unsigned int* counters, const void* exp_src, void* exp_dst,
                                   size_t exp_bytes, const unsigned int* rtab, const unsigned int* escapes, PhaseParams* pp, hipStream_t stream) {
  const int vr = k4s_vr(capval);   // pp != NULL: capval is the caller's guess (it fixes vr and SW7), ncompact an upper bound
  const int grid = k4s_grid(ncompact);
  const size_t lds = (size_t)vr * (vr <= 128 ? kS4Cols : kResClasses) * 4;
  unsigned int* sl = static_cast<unsigned int*>(slabs);
  const int pg = fold_per_group_add(grid);
  const bool sw7 = capval <= 127;
#define RSI_K4S(C64, RW) do { RSI_ALLOW_FULL_LDS((k_rescale_compact_stream<C64, RW>));                                                                   \
    RSI_LAUNCH((k_rescale_compact_stream<C64, RW>), dim3(grid), dim3(kS4Threads), lds, stream, depth8, depth, gcbits, n, n / 64 + 1, table, cbreak, cum, \
               nreg, ncompact, capval, m, vr, sw7 ? 1 : 0, rdc, res_hist, sl, pg, counters, exp_src, exp_dst, (unsigned int)exp_bytes, inl, rtab, escapes, pp); } while (0)
  const bool rawmode = rtab == nullptr;   // the bytes are final values (the caller has no ratios to hand over: -NOGC, the three-pass chain)
  if (vr <= 128) { if (rawmode) RSI_K4S(true, true); else RSI_K4S(true, false); }
  else { if (rawmode) RSI_K4S(false, true); else RSI_K4S(false, false); }
#undef RSI_K4S
}

// the bins' medians and sums from the bytes the launch above leaves: PARTS lanes per bin, seven dwords each
void launch_bin_median8(const uint8_t* rdc, int64_t ncompact, int32_t capval, int m, int32_t* binmed, int64_t* binsum, const PhaseParams* pp,
                        hipStream_t stream) {
  const int vr = k4s_vr(capval);
  const bool sw7 = capval <= 127;
  const int parts = m <= 52 ? 2 : (m <= 104 ? 4 : (m <= 216 ? 8 : 16));
  const int64_t nb = ncompact / m;
  const int64_t ntrips = (nb + 64 / parts - 1) / (64 / parts);
  int64_t mg = (ntrips + kM4Threads / 64 - 1) / (kM4Threads / 64);
  mg = mg < 1 ? 1 : (mg > kM4Grid ? kM4Grid : mg);
#define RSI_K4M(SW, PT) RSI_LAUNCH((k_bin_median8<SW, PT>), dim3((unsigned)mg), dim3(kM4Threads), 0, stream, rdc, ncompact, capval, m, vr, binmed, binsum, pp)
  if (sw7) { if (parts == 2) RSI_K4M(true, 2); else if (parts == 4) RSI_K4M(true, 4); else if (parts == 8) RSI_K4M(true, 8); else RSI_K4M(true, 16); }
  else { if (parts == 2) RSI_K4M(false, 2); else if (parts == 4) RSI_K4M(false, 4); else if (parts == 8) RSI_K4M(false, 8); else RSI_K4M(false, 16); }
#undef RSI_K4M
}

}  // namespace rsik
