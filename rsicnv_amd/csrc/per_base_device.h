// per_base_device.h -- small device-side helpers shared by the per-base kernel files (kernels_base.hip, kernels_k4s.hip):
// byte tests, quad sums without the LDS pipeline, window GC counts on staged mask words, the removed-region table,
// the float form of the GC rescale.  Each translation unit gets its own copy (anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"
#include "device_util.h"

#if defined(__HIPCC__)
namespace rsik {
namespace {

constexpr int kFixShift = 22;   // fraction bits of the fixed-point ratios K2j verifies and K4j / K4s rescale with

__device__ inline int lane_id() { return threadIdx.x & 63; }
__device__ inline bool has_escape(uint32_t w) {   // any byte of w equal to 0xff
  const uint32_t x = ~w;                          // a zero byte of x
  return ((x - 0x01010101u) & ~x & 0x80808080u) != 0;
}


// Sum over the 2, 4, 8 or 16 consecutive lanes that share a bin (the median phases of K4' / K4j): neighbours at distance 1 and 2
// through DPP quad permutes -- one VALU instruction each -- instead of ds_bpermute, which goes through the LDS pipeline and whose
// latency sat seven times two deep in every bin's bisection.
__device__ inline int dpp_xor1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); }   // quad_perm [1, 0, 3, 2]
__device__ inline int dpp_xor2(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); }   // quad_perm [2, 3, 0, 1]
__device__ inline int parts_sum(int x, int parts) {
  x += dpp_xor1(x);
  if (parts > 2) x += dpp_xor2(x);
  for (int d = 4; d < parts; d <<= 1) x += __shfl_xor(x, d);
  return x;
}
__device__ inline uint64_t dpp_xor1_u64(uint64_t x) { return (uint64_t)(uint32_t)dpp_xor1((int)(uint32_t)x) | ((uint64_t)(uint32_t)dpp_xor1((int)(uint32_t)(x >> 32)) << 32); }
__device__ inline uint64_t dpp_xor2_u64(uint64_t x) { return (uint64_t)(uint32_t)dpp_xor2((int)(uint32_t)x) | ((uint64_t)(uint32_t)dpp_xor2((int)(uint32_t)(x >> 32)) << 32); }

// the same two queries on a plain array of staged words (a workgroup's tile in K4j)
__device__ inline uint32_t gcw_window(const uint64_t* __restrict__ word, uint32_t rel) {
  const uint32_t k = rel >> 6, b = rel & 63;
  const uint64_t w0 = word[k], w1 = word[k + 1], w2 = word[k + 2], w3 = word[k + 3], w4 = word[k + 4];
  const uint32_t rem = 9 + b;
  const uint64_t m3 = rem >= 64 ? ~0ull : ((1ull << rem) - 1);
  const uint64_t m4 = rem > 64 ? ((1ull << (rem - 64)) - 1) : 0ull;
  return (uint32_t)(__popcll(w0 >> b) + __popcll(w1) + __popcll(w2) + __popcll(w3 & m3) + __popcll(w4 & m4));
}
__device__ inline uint32_t gcw_field16(const uint64_t* __restrict__ word, uint32_t rel) {
  const uint32_t k = rel >> 6, b = rel & 63;
  uint64_t v = word[k] >> b;
  if (b > 48) v |= word[k + 1] << (64 - b);
  return (uint32_t)v & 0xffffu;
}

// The GC rescale without a division -- or any double arithmetic -- per base (f64 runs at half rate on gfx950, its
// conversions at a quarter).  The reference's expression is (int)((double)d * rdmean / table[g] + 0.5) (gccontent.cpp:89:
// product, IEEE division, truncation).  With ratio = (float)(rdmean / table[g]), f = fma((float)d, ratio, 0.5f) is within
// 2^-23 (t + 0.5) of the reference's t + 0.5 (one rounding in the ratio, one in the fma; d < 2^24 is exact), so
// floor(f) equals the reference's result unless f lies within tol = 2.5e-7 f + 1e-6 (twice that bound) of an integer.
// Those bases raise `unsure` and the caller redoes them with the reference's own double expression: the result is the
// reference's in every case; only one base in ten thousand takes the slow way.
__device__ inline float rescale_f32(float d, float ratio, bool& unsure) {
  const float f = __fmaf_rn(d, ratio, 0.5f);
  const float fl = floorf(f);
  const float fr = f - fl;
  const float tol = __fmaf_rn(f, 2.5e-7f, 1.0e-6f);
  unsure |= fabsf(fr - 0.5f) > 0.5f - tol;
  return fl;
}

constexpr int kRegLds = 128;   // removed regions mirrored in LDS (the list is short; more stay in HBM; 128: K4j keeps four workgroups per CU with its 24 KB histogram)

struct RegionTable {
  const int64_t* cbreak; const int64_t* cum; int nreg;
  int64_t* s_break; int64_t* s_cum;   // LDS mirror of the first kRegLds entries (+1 for cum)
  __device__ int64_t brk(int k) const { return k < kRegLds ? s_break[k] : cbreak[k]; }
  __device__ int64_t shift(int k) const { return k <= kRegLds ? s_cum[k] : cum[k]; }
};

// #GC in [lo, lo + 201) straight from the mask in HBM (per-element path only)
__device__ inline int gc_count201(const uint64_t* __restrict__ gcbits, int64_t lo) {
  int c = 0;
  int64_t p = lo;
  const int64_t end = lo + 201;
  while (p < end) {
    const int64_t w = p >> 6;
    const int b = (int)(p & 63);
    int take = 64 - b;
    if (p + take > end) take = (int)(end - p);
    uint64_t x = gcbits[w] >> b;
    if (take < 64) x &= (1ull << take) - 1;
    c += __popcll(x);
    p += take;
  }
  return c;
}

struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t x, y, z, w; };   // 16 bytes at any byte address (gfx950 loads them in one go)

}  // namespace
}  // namespace rsik
#endif
