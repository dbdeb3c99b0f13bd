// kernels_io.hip -- depth text ingestion on the device.  The reference's loader
// (load_data_from_text, loaddata.cpp:496-517) reads "pos depth" lines with getline + istringstream:
// about 65 % of its wall time on the -d path.  Here the file's bytes are streamed to HBM in pinned
// chunks and parsed by one kernel at memory speed.
//
// Reference semantics kept: empty lines and lines starting with '#' are skipped; `iss >> pos >> d`
// (leading blanks, optional sign, digits; a failed extraction leaves 0); pos < 1 skipped; reading
// STOPS at the first pos >= size (the last base is never set, App. A Q7); RD[pos-1] = d, later lines
// overwrite earlier ones.  The last two rules are order-dependent, so the kernel also proves that
// positions are strictly increasing through the file (true for any samtools-depth style file): then
// "stop at the first pos >= size" equals "ignore every pos >= size" and no position is written twice.
// When the proof fails the caller falls back to the sequential host parser (same semantics, slower).
#include "kernels.h"

namespace rsik {

namespace {

constexpr int kThreads = 256;
constexpr int kSpan = 32;                      // bytes of text per thread: the lines that START in them are the thread's
constexpr int kTile = kThreads * kSpan;        // 8 KB of text per workgroup

__device__ inline bool is_blank(unsigned char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// `iss >> v` on [q, e): leading blanks, optional sign, digits.  false (v = 0) when no digit follows.
__device__ inline bool parse_int(const unsigned char* __restrict__ t, long long& q, long long e, long long& v) {
  while (q < e && is_blank(t[q])) ++q;
  bool neg = false;
  if (q < e && (t[q] == '-' || t[q] == '+')) { neg = t[q] == '-'; ++q; }
  if (q >= e || t[q] < '0' || t[q] > '9') { v = 0; return false; }
  long long x = 0;
  while (q < e && t[q] >= '0' && t[q] <= '9') { x = x * 10 + (t[q] - '0'); ++q; }
  v = neg ? -x : x;
  return true;
}

__global__ __launch_bounds__(kThreads) void k_parse_depth_text(const unsigned char* __restrict__ text, long long nbytes,
                                                              long long size, int32_t* __restrict__ depth,
                                                              long long* __restrict__ wg_first, long long* __restrict__ wg_max,
                                                              TextParseStats* __restrict__ stats) {
  __shared__ long long s_first[kThreads], s_max[kThreads];
  __shared__ int s_bad;
  if (threadIdx.x == 0) s_bad = 0;
  const long long b0 = ((long long)blockIdx.x * kThreads + threadIdx.x) * kSpan;
  long long first = -1, last = -1;     // positions of the thread's first / latest counted line (pos >= 1)
  unsigned lines = 0, stored = 0, beyond = 0;
  bool sorted = true;
  if (b0 < nbytes) {
    const long long b1 = b0 + kSpan < nbytes ? b0 + kSpan : nbytes;
    for (long long s = b0; s < b1; ++s) {
      if (s != 0 && text[s - 1] != '\n') continue;          // not a line start (the chunk itself starts on one)
      long long e = s;
      while (e < nbytes && text[e] != '\n') ++e;              // lines are short; they may run past the span
      if (e == s || text[s] == '#') continue;
      long long q = s, pos = 0, d = 0;
      if (!parse_int(text, q, e, pos)) continue;              // extraction failed: pos = 0, skipped below anyway
      parse_int(text, q, e, d);
      if (pos < 1) continue;
      ++lines;
      if (last >= 0 && pos <= last) sorted = false;
      if (first < 0) first = pos;
      last = pos;
      if (pos >= size) { ++beyond; continue; }
      depth[pos - 1] = (int32_t)d;
      ++stored;
    }
  }
  // ---- strictly increasing across the threads of the workgroup: running maximum of `last` ----
  s_first[threadIdx.x] = first;
  s_max[threadIdx.x] = last;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long run = -1, wfirst = -1;
    bool ok = true;
    for (int t = 0; t < kThreads; ++t) {
      if (s_first[t] < 0) continue;
      if (wfirst < 0) wfirst = s_first[t];
      if (run >= 0 && s_first[t] <= run) ok = false;
      run = s_max[t] > run ? s_max[t] : run;
    }
    wg_first[blockIdx.x] = wfirst;
    wg_max[blockIdx.x] = run;
    if (!ok) s_bad = 1;
  }
  if (!sorted) atomicOr(&s_bad, 1);
  __syncthreads();
  // ---- totals ----
  for (int d = 32; d >= 1; d >>= 1) { lines += __shfl_xor(lines, d); stored += __shfl_xor(stored, d); beyond += __shfl_xor(beyond, d); }
  // one set of atomics per workgroup (they all land on the same three words and serialise there)
  __shared__ int s_tot[kThreads / 64][3];
  if ((threadIdx.x & 63) == 0) { s_tot[threadIdx.x >> 6][0] = (int)lines; s_tot[threadIdx.x >> 6][1] = (int)stored; s_tot[threadIdx.x >> 6][2] = (int)beyond; }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t0 = 0, t1 = 0, t2 = 0;
    for (int w = 0; w < kThreads / 64; ++w) { t0 += s_tot[w][0]; t1 += s_tot[w][1]; t2 += s_tot[w][2]; }
    if (t0) atomicAdd(&stats->lines, (unsigned long long)t0);
    if (t1) atomicAdd(&stats->stored, (unsigned long long)t1);
    if (t2) atomicAdd(&stats->beyond, (unsigned long long)t2);
    if (s_bad) atomicOr(&stats->unsorted, 1u);
  }
}

}  // namespace

int text_parse_workgroups(long long nbytes) { return (int)((nbytes + kTile - 1) / kTile); }

void launch_parse_depth_text(const void* text, long long nbytes, long long size, int32_t* depth, long long* wg_first,
                             long long* wg_max, TextParseStats* stats, hipStream_t stream) {
  const int grid = text_parse_workgroups(nbytes);
  if (grid <= 0) return;
  RSI_LAUNCH(k_parse_depth_text, dim3(grid), dim3(kThreads), 0, stream, static_cast<const unsigned char*>(text), nbytes, size,
                     depth, wg_first, wg_max, stats);
}

}  // namespace rsik

// ------------------------------------------------------------------------------------------
// BAM pileup -> per-base depth.  The host inflates the BGZF blocks and finds the
// record boundaries; everything the reference does per read -- the filters of load_data_from_bam
// (loaddata.cpp:313-320), the CIGAR positions of resolve_cigar_pos (samfunctions.cpp:38-100) and the
// per-base quality test (loaddata.cpp:328-331) -- runs here, one thread per record, on the inflated
// bytes in HBM.  A run of counted bases becomes +1 / -1 in a difference array; an inclusive scan turns
// that into the depth.
namespace rsik {

namespace {

__device__ inline uint32_t ld_u32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ inline uint32_t ld_u16(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

enum { kCigM = 0, kCigI = 1, kCigD = 2, kCigN = 3, kCigS = 4, kCigEq = 7, kCigX = 8 };

__global__ __launch_bounds__(256) void k_bam_depth(const unsigned char* __restrict__ data, const uint32_t* __restrict__ rec_off,
                                                   int nrec, int tid, int minq, int min_baseq, long long n,
                                                   int32_t* __restrict__ diff, BamDepthStats* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned used = 0, runs = 0, bad = 0;
  if (i < nrec) {
    const long long bsize = (long long)ld_u32(data + rec_off[i]);   // block_size: bytes of the record behind this field
    const unsigned char* b = data + rec_off[i] + 4;       // past block_size
    const int rtid = (int)ld_u32(b), pos0 = (int)ld_u32(b + 4);
    const int l_name = b[8], mapq = b[9];
    const int n_cig = (int)ld_u16(b + 12), flag = (int)ld_u16(b + 14);
    const int l_seq = (int)ld_u32(b + 16);
    // A record whose variable-length fields do not fit its block_size (corrupt or crafted file) is counted and skipped:
    // nothing below may read past the record or write outside the depth array.  pos < 0 (an unplaced read that carries a
    // reference id) is never yielded by the reference's region iterator: skipped like the other filtered reads.
    const bool fits = l_seq >= 0 && 32ll + l_name + 4ll * n_cig + ((long long)l_seq + 1) / 2 + (long long)l_seq <= bsize;
    if (!fits) bad = 1;
    // loaddata.cpp:315-319: pos == 0, tid < 0, mapq, secondary, duplicate (the iterator only yields this tid)
    const bool keep = fits && rtid == tid && pos0 > 0 && mapq >= minq && !(flag & 0x100) && !(flag & 0x400);
    if (keep) {
      const unsigned char* cig = b + 32 + l_name;
      const unsigned char* qual = cig + 4 * n_cig + (l_seq + 1) / 2;
      // anchor: the first M / D / = / X (samfunctions.cpp:75-78); without one the read contributes nothing
      int anchor = -1;
      for (int k = 0; k < n_cig && anchor < 0; ++k) { const int op = cig[4 * k] & 0xf; if (op == kCigM || op == kCigD || op == kCigEq || op == kCigX) anchor = k; }
      if (anchor >= 0) {
        used = 1;
        long long ref_end = (long long)pos0 + 1;     // 1-based position of op `anchor` (samfunctions.cpp:85-92)
        int q = 0;
        for (int k = 0; k < n_cig; ++k) {
          const uint32_t c = ld_u32(cig + 4 * k);
          const int op = (int)(c & 0xf), len = (int)(c >> 4);
          if (k >= anchor && (op == kCigM || op == kCigEq)) {
            const long long p1 = ref_end - 1;        // 0-based reference position of the op's first base
            // bases with quality >= min_baseq, in runs (loaddata.cpp:328-331)
            long long run_start = -1;
            for (int t = 0; t < len; ++t) {
              const long long p = p1 + t;
              if (p >= n || q + t >= l_seq) break;   // the second: a CIGAR longer than the read (malformed)
              const bool ok = qual[q + t] >= min_baseq;
              if (ok && run_start < 0) run_start = p;
              if (!ok && run_start >= 0) { atomicAdd(&diff[run_start], 1); atomicAdd(&diff[p], -1); run_start = -1; ++runs; }
            }
            if (run_start >= 0) {
              long long e = p1 + len; if (e > n) e = n;
              if (q + len > l_seq) { const long long eq = p1 + (l_seq - q); e = eq < e ? eq : e; }
              if (e > run_start) { atomicAdd(&diff[run_start], 1); atomicAdd(&diff[e], -1); ++runs; }
            }
          }
          // ops before the anchor never match (they are I / S / H / N / P); their positions are not needed
          if (k >= anchor && (op == kCigM || op == kCigD || op == kCigN || op == kCigS)) ref_end += len;   // '=' and 'X' do not advance: reference quirk
          if (op == kCigM || op == kCigI || op == kCigS || op == kCigEq || op == kCigX) q += len;          // samfunctions.cpp:67-71
        }
      }
    }
  }
  for (int d = 32; d >= 1; d >>= 1) { used += __shfl_xor(used, d); runs += __shfl_xor(runs, d); bad += __shfl_xor(bad, d); }
  if ((threadIdx.x & 63) == 0) {
    if (used) atomicAdd(&stats->used, (unsigned long long)used);
    if (runs) atomicAdd(&stats->runs, (unsigned long long)runs);
    if (bad) atomicAdd(&stats->malformed, (unsigned long long)bad);
  }
}

// ---- inclusive scan of int32 in place: tile sums, scan of the tile sums, tile scan + offset ----
constexpr int kScanThreads = 256, kScanPer = 16, kScanTileElems = kScanThreads * kScanPer;

__device__ inline int wg_exscan_i32(int v, int* s_w /* 4 */, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
  for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
  __syncthreads();
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < kScanThreads / 64; ++w) { if (w < wave) base += s_w[w]; tot += s_w[w]; }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void k_scan_tile_sums(const int32_t* __restrict__ x, long long n, int32_t* __restrict__ tile_sum) {
  __shared__ int s_w[kScanThreads / 64];
  const long long e0 = (long long)blockIdx.x * kScanTileElems + (long long)threadIdx.x * kScanPer;
  int s = 0;
  for (int k = 0; k < kScanPer; ++k) if (e0 + k < n) s += x[e0 + k];
  int total;
  (void)wg_exscan_i32(s, s_w, &total);
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanThreads) void k_scan_tile_offsets(int32_t* __restrict__ tile_sum, int ntiles) {   // exclusive, in place, one workgroup
  __shared__ int s_w[kScanThreads / 64];
  int carry = 0;
  for (int t0 = 0; t0 < ntiles; t0 += kScanThreads) {
    const int t = t0 + (int)threadIdx.x;
    const int v = t < ntiles ? tile_sum[t] : 0;
    int total;
    const int ex = wg_exscan_i32(v, s_w, &total);
    if (t < ntiles) tile_sum[t] = carry + ex;
    carry += total;
  }
}
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(int32_t* __restrict__ x, long long n, const int32_t* __restrict__ tile_off) {
  __shared__ int s_w[kScanThreads / 64];
  const long long e0 = (long long)blockIdx.x * kScanTileElems + (long long)threadIdx.x * kScanPer;
  int v[kScanPer];
  int s = 0;
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) { v[k] = e0 + k < n ? x[e0 + k] : 0; s += v[k]; v[k] = s; }
  int total;
  const int base = tile_off[blockIdx.x] + wg_exscan_i32(s, s_w, &total);
#pragma unroll
  for (int k = 0; k < kScanPer; ++k) if (e0 + k < n) x[e0 + k] = base + v[k];
}

}  // namespace

void launch_bam_depth(const void* data, const uint32_t* rec_off, int nrec, int tid, int minq, int min_baseq, long long n, int32_t* diff,
                      BamDepthStats* stats, hipStream_t stream) {
  if (nrec <= 0) return;
  RSI_LAUNCH(k_bam_depth, dim3((nrec + 255) / 256), dim3(256), 0, stream, static_cast<const unsigned char*>(data), rec_off, nrec, tid,
                     minq, min_baseq, n, diff, stats);
}
int scan_tiles(long long n) { return (int)((n + kScanTileElems - 1) / kScanTileElems); }
void launch_inclusive_scan_i32(int32_t* x, long long n, int32_t* tile_scratch, hipStream_t stream) {
  const int ntiles = scan_tiles(n);
  if (ntiles <= 0) return;
  RSI_LAUNCH(k_scan_tile_sums, dim3(ntiles), dim3(kScanThreads), 0, stream, x, n, tile_scratch);
  RSI_LAUNCH(k_scan_tile_offsets, dim3(1), dim3(kScanThreads), 0, stream, tile_scratch, ntiles);
  RSI_LAUNCH(k_scan_apply, dim3(ntiles), dim3(kScanThreads), 0, stream, x, n, tile_scratch);
}

}  // namespace rsik
