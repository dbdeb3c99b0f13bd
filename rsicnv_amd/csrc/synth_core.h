// synth_core.h -- counter-based synthetic chromosome generator shared by the host (g++) and
// device (hipcc, gfx950) builds.  Every base is a pure function of (seed, position, small tables),
// so the host and the GPU produce bit-identical FASTA bytes and depth values in any order.
// Only integer arithmetic and table look-ups happen per base; the tables themselves are built on
// the host with IEEE + - * / only (no libm), see synth_tables.cpp.
//
// This is input generation for tests/bench, not part of the hot path.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RSI_HD __host__ __device__ inline
#else
#define RSI_HD inline
#endif

// One interval [beg, end) with an integer code (copy-number class, or unused).
struct rsi_synth_interval {
  int64_t beg;
  int64_t end;
  int32_t code;
  int32_t pad;
};

// Copy-number classes used by implanted events.
enum { RSI_CN_1X = 0, RSI_CN_0X = 1, RSI_CN_HALF = 2, RSI_CN_1P5 = 3, RSI_CN_2X = 4, RSI_CN_CLASSES = 5 };

#define RSI_SYNTH_GC_LEVELS 202   /* window GC count 0..201 */
#define RSI_SYNTH_WAVE 1024       /* entries in the GC-probability wave */
#define RSI_SYNTH_WAVE_STEP 137   /* bases per wave entry: period 1024*137 = 140,288 bp */

RSI_HD uint64_t rsi_mix64(uint64_t z) {   // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// i-th output of the splitmix64 stream started at `seed` (random access).
RSI_HD uint64_t rsi_stream(uint64_t seed, uint64_t i) {
  return rsi_mix64(seed + (i + 1) * 0x9E3779B97F4A7C15ULL);
}

// Binary search in sorted disjoint intervals; returns index or -1.
RSI_HD int rsi_find_interval(const rsi_synth_interval* iv, int n, int64_t pos) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    if (pos < iv[mid].beg) hi = mid - 1;
    else if (pos >= iv[mid].end) lo = mid + 1;
    else return mid;
  }
  return -1;
}

// FASTA byte at position i.  wave[k] = GC probability * 2^32 (uint32).
RSI_HD uint8_t rsi_synth_base(uint64_t seed_fa, int64_t i, const uint32_t* wave,
                              const rsi_synth_interval* nruns, int n_nruns,
                              const rsi_synth_interval* lower, int n_lower) {
  if (rsi_find_interval(nruns, n_nruns, i) >= 0) return (uint8_t)'N';
  uint64_t h = rsi_stream(seed_fa, (uint64_t)i);
  uint32_t u = (uint32_t)(h >> 32);
  uint32_t thr = wave[(i / RSI_SYNTH_WAVE_STEP) % RSI_SYNTH_WAVE];
  int bit = (int)(h & 1);
  uint8_t c = (u < thr) ? (bit ? 'G' : 'C') : (bit ? 'A' : 'T');
  if (rsi_find_interval(lower, n_lower, i) >= 0) c = (uint8_t)(c + 32);   // soft-masked
  return c;
}

RSI_HD int rsi_is_gc(uint8_t c) { return c == 'G' || c == 'C'; }

// Inverse-CDF sample: number of thresholds <= u.  thr[0..len) ascending, u in [0, 2^53).
RSI_HD int32_t rsi_synth_sample(const uint64_t* thr, int len, uint64_t u) {
  int lo = 0, hi = len;   // first index with thr[idx] > u
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (thr[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Depth at position i.  cls = gc_level * RSI_CN_CLASSES + cn (gc_level = 0 for the Poisson model).
RSI_HD int32_t rsi_synth_depth(uint64_t seed_rd, int64_t i, int cls, const uint64_t* thr_all,
                               const int32_t* thr_off) {
  uint64_t u = rsi_stream(seed_rd, (uint64_t)i) >> 11;
  int off = thr_off[cls];
  int len = thr_off[cls + 1] - off;
  return rsi_synth_sample(thr_all + off, len, u);
}
