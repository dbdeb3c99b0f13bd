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
  if ((threadIdx.x & 63) == 0) {
    if (lines) atomicAdd(&stats->lines, (unsigned long long)lines);
    if (stored) atomicAdd(&stats->stored, (unsigned long long)stored);
    if (beyond) atomicAdd(&stats->beyond, (unsigned long long)beyond);
  }
  if (threadIdx.x == 0 && s_bad) atomicOr(&stats->unsorted, 1u);
}

}  // namespace

int text_parse_workgroups(long long nbytes) { return (int)((nbytes + kTile - 1) / kTile); }

void launch_parse_depth_text(const void* text, long long nbytes, long long size, int32_t* depth, long long* wg_first,
                             long long* wg_max, TextParseStats* stats, hipStream_t stream) {
  const int grid = text_parse_workgroups(nbytes);
  if (grid <= 0) return;
  hipLaunchKernelGGL(k_parse_depth_text, dim3(grid), dim3(kThreads), 0, stream, static_cast<const unsigned char*>(text), nbytes, size,
                     depth, wg_first, wg_max, stats);
}

}  // namespace rsik
