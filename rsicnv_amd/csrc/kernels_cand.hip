// kernels_cand.hip -- candidate stages on the device: boundary refinement
// (optimize_with_derivative, rsi.cpp:889-944) and the neighbourhood test (isitcnvwrap + isitcnv,
// rsi.cpp:175-287, 101-172) directly on the compacted depth in HBM, one workgroup per candidate.
// The host keeps the list logic (who is whose neighbour, what is deleted, merge decisions); the
// device does every array-sized step, so the per-base depth never has to be paged to the host.
//
// Exactness: all sums of depths are integers (int64); window means are (float)((double)S/width)
// as in the reference; quantiles come from integer histograms on the reference's grids.  The one
// deviation is the second moment of the window means, summed here as a tree of doubles instead of
// sequentially (relative difference ~1e-15; it only enters the p-value).
#include "kernels.h"

namespace rsik {

namespace {

constexpr int kThreads = 256;        // boundary refinement, range sums
constexpr int kTestThreads = 1024;   // neighbourhood test: one workgroup per candidate, so make it a big one
constexpr int kMaxWaves = kTestThreads / 64;
__device__ inline int lane_id() { return threadIdx.x & 63; }

// ---- block-wide helpers (256 threads) --------------------------------------------------------
template <class T, class Op>
__device__ inline T block_reduce(T v, Op op, T* s_tmp /* kMaxWaves */) {
  for (int d = 32; d >= 1; d >>= 1) v = op(v, __shfl_xor(v, d));
  __syncthreads();
  if (lane_id() == 0) s_tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  T r = s_tmp[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = op(r, s_tmp[w]);
  __syncthreads();
  return r;
}
// exclusive prefix of one int per thread; returns the thread's offset, *total = sum over the block
__device__ inline int block_exscan(int v, int* s_tmp /* kMaxWaves */, int* total) {
  const int incl = wave_incl_scan(v);
  __syncthreads();
  if (lane_id() == 63) s_tmp[threadIdx.x >> 6] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { if (w < (int)(threadIdx.x >> 6)) base += s_tmp[w]; tot += s_tmp[w]; }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}
// exact int64 prefix of f(0..len-1) into P[0..len] (P[0] = 0), in tiles of blockDim.x consecutive elements
// (coalesced reads and writes) with a running carry
template <class F>
__device__ inline void block_prefix_i64(long long* __restrict__ P, int len, F f, long long* s_tmp /* kMaxWaves */);

// exclusive prefix of one int64 per thread over the block; *total = block sum
__device__ inline long long block_exscan_i64(long long v, long long* s_tmp /* kMaxWaves */, long long* total) {
  const long long incl = wave_incl_scan(v);
  __syncthreads();
  if (lane_id() == 63) s_tmp[threadIdx.x >> 6] = incl;
  __syncthreads();
  long long base = 0, tot = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { if (w < (int)(threadIdx.x >> 6)) base += s_tmp[w]; tot += s_tmp[w]; }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

template <class F>
__device__ inline void block_prefix_i64(long long* __restrict__ P, int len, F f, long long* s_tmp) {
  long long carry = 0;
  if (threadIdx.x == 0) P[0] = 0;
  for (int t0 = 0; t0 < len; t0 += (int)blockDim.x) {
    const int e = t0 + (int)threadIdx.x;
    const long long v = e < len ? f(e) : 0;
    long long total;
    const long long ex = block_exscan_i64(v, s_tmp, &total);
    if (e < len) P[e + 1] = carry + ex + v;
    carry += total;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Boundary refinement (optimize_with_derivative, rsi.cpp:889-944; detectcnv calls it twice, rsi.cpp:
// 1876-1877: one launch per call).  dd[i] = sum of the len values left of q = from+i minus the sum of
// the len values from q on; the new start is the first maximum (DEL) / minimum (DUP) of dd over its
// first 2*reach entries, the new end the first minimum / maximum over its last 2*reach entries; only
// strictly positive / negative values count and index 0 never moves anything.
// Each of the two search windows of a candidate is cut into kEdgeChunks pieces, one workgroup each.
// A workgroup walks its piece in tiles of 256 with dd[i+1] - dd[i] = -p[q-len] + 2 p[q] - p[q+len] and an
// exact int64 block scan, RELATIVE to the piece's first index: the extremum of a piece sits at the same index
// whatever the offset, and the sign test is monotone in the value, so both wait for the fold.  dd at a
// window's first index (a sum over 2*len values) is itself shared out: every workgroup of the window adds
// a sixteenth of it to the window's accumulator.  The last workgroup of a candidate to finish chains the
// pieces (offset of piece c = window start + totals of the pieces before it), applies the sign test, folds
// (smallest index wins ties, as the reference's strict comparisons do) and writes the new coordinates.
// Per candidate this reads about 7*len values instead of 2*len per workgroup.
constexpr int kEdgeChunks = 16;
struct ArgBest { long long v; int i; };
__device__ inline ArgBest arg_pick(ArgBest a, ArgBest b, bool want_max) {
  if (b.i < 0) return a;
  if (a.i < 0) return b;
  if (a.v != b.v) return (want_max ? (b.v > a.v) : (b.v < a.v)) ? b : a;
  return b.i < a.i ? b : a;
}

struct SharpenWs {   // views into the workspace of one launch (sharpen_workspace_bytes)
  unsigned long long* start_acc;   // [njobs][2]   dd at the first index of each window, accumulated; zero between launches
  uint32_t* done;                  // [njobs]      finished workgroups; zero between launches
  long long* part_v;               // [njobs][32]  extremum of the piece, relative to its first index
  long long* part_t;               // [njobs][32]  sum of the piece's steps
  int32_t* part_i;                 // [njobs][32]  index of the extremum (-1: empty piece)
};
__host__ __device__ inline size_t sharpen_zero_bytes(int njobs) { return ((size_t)njobs * 16 + (size_t)njobs * 4 + 15) & ~size_t(15); }
__host__ __device__ inline SharpenWs sharpen_views(void* ws, int njobs) {
  unsigned char* p = static_cast<unsigned char*>(ws);
  const size_t slots = (size_t)njobs * 2 * kEdgeChunks;
  SharpenWs w;
  w.start_acc = reinterpret_cast<unsigned long long*>(p);
  w.done = reinterpret_cast<uint32_t*>(p + (size_t)njobs * 16);
  p += sharpen_zero_bytes(njobs);
  w.part_v = reinterpret_cast<long long*>(p);
  w.part_t = w.part_v + slots;
  w.part_i = reinterpret_cast<int32_t*>(w.part_t + slots);
  return w;
}

template <typename TD>
__global__ __launch_bounds__(kThreads) void k_sharpen_edges(const TD* __restrict__ rdc, int64_t ncompact,
                                                            EdgeJob* __restrict__ jobs, int njobs /* capacity the workspace is laid out for */,
                                                            void* __restrict__ ws) {
  __shared__ long long s_l[kMaxWaves];
  __shared__ long long s_v[kThreads / 64];
  __shared__ int s_i[kThreads / 64];
  __shared__ int s_last;
  const SharpenWs W = sharpen_views(ws, njobs);
  const int jb = blockIdx.y;
  const EdgeJob job = jobs[jb];
  const int len = job.end - job.start + 1;
  const int reach = len / 4 > 250 ? len / 4 : 250;
  const int from = job.start - reach, to = job.end + reach;
  if (job.type > 1) return;                                                  // untyped candidate: the reference moves nothing
  if (from < 2 * len || (int64_t)to > ncompact - 2 * (int64_t)len) return;   // too close to the ends: unchanged
  const int nstep = to - from;
  const int win = blockIdx.x / kEdgeChunks, chunk = blockIdx.x % kEdgeChunks;   // win 0: start search, 1: end search
  const int wlen = 2 * reach;
  // The two search windows are the first and the last 2 * reach entries of dd[0 .. nstep).  A candidate whose first
  // refinement left end < start (len <= 0) has FEWER than 2 * reach entries: the reference's loops then index its vector out
  // of range (rsi.cpp:917, 930: undefined); here, as in the host path (host_calls.cpp:sharpen_edges) and the oracle, both
  // searches cover the entries that exist.  (Until round 4 this kernel continued dd's recurrence beyond the vector instead --
  // a third answer, found by tests/test_fallback_paths.py's bimodal chromosome as one call's neighbourhood statistics.)
  const int wbase = win == 0 ? 0 : (nstep - wlen > 0 ? nstep - wlen : 0);
  const int wend = win == 0 ? (wlen < nstep ? wlen : nstep) : nstep;
  const int wspan = wend > wbase ? wend - wbase : 0;
  const int cs = (wspan + kEdgeChunks - 1) / kEdgeChunks;
  const int i0 = wbase + chunk * cs;
  int i1 = i0 + cs; if (i1 > wend) i1 = wend;
  const bool del = job.type == 0;
  const bool want_max = win == 0 ? del : !del;
  {   // this workgroup's share of dd at the window's first index
    const int q0 = from + wbase;
    const int js = (len + kEdgeChunks - 1) / kEdgeChunks;
    const int j0 = chunk * js;
    int j1 = j0 + js; if (j1 > len) j1 = len;
    long long acc = 0;
    for (int j = j0 + (int)threadIdx.x; j < j1; j += kThreads) acc += (long long)rdc[q0 - len + j] - (long long)rdc[q0 + j];
    acc = block_reduce(acc, [](long long a, long long b) { return a + b; }, s_l);
    if (threadIdx.x == 0 && acc != 0) atomicAdd(&W.start_acc[jb * 2 + win], (unsigned long long)acc);
  }
  ArgBest best{0, -1};
  long long cur = 0;   // dd relative to the piece's first index
  for (int t0 = i0; t0 < i1; t0 += kThreads) {
    const int i = t0 + (int)threadIdx.x;
    long long delta = 0;
    if (i < i1) { const int q = from + i; delta = -(long long)rdc[q - len] + 2ll * rdc[q] - (long long)rdc[q + len]; }
    long long total;
    const long long dd = cur + block_exscan_i64(delta, s_l, &total);
    if (i < i1) best = arg_pick(best, ArgBest{dd, i}, want_max);
    cur += total;
  }
  // block arg-reduction (values first, then smallest index)
  for (int d = 32; d >= 1; d >>= 1) { ArgBest o; o.v = __shfl_xor(best.v, d); o.i = __shfl_xor(best.i, d); best = arg_pick(best, o, want_max); }
  __syncthreads();
  if (lane_id() == 0) { s_v[threadIdx.x >> 6] = best.v; s_i[threadIdx.x >> 6] = best.i; }
  __syncthreads();
  if (threadIdx.x == 0) {
    ArgBest r{s_v[0], s_i[0]};
    for (int w = 1; w < kThreads / 64; ++w) r = arg_pick(r, ArgBest{s_v[w], s_i[w]}, want_max);
    const size_t slot = (size_t)jb * 2 * kEdgeChunks + blockIdx.x;
    // hand-over without fences (device_util.h): write-through stores, wait for them, then the arrival counter; the reader
    // uses coherent loads.  An agent-scope fence here writes back the XCD's whole L2 -- in a pool that is other
    // chromosomes' streaming output, and 1280 workgroups per launch each waited for it.
    st_cg(reinterpret_cast<unsigned long long*>(&W.part_v[slot]), (unsigned long long)r.v);
    st_cg(reinterpret_cast<unsigned int*>(&W.part_i[slot]), (unsigned int)r.i);
    st_cg(reinterpret_cast<unsigned long long*>(&W.part_t[slot]), (unsigned long long)cur);
    drain();
    s_last = atomicAdd(&W.done[jb], 1u) == 2 * kEdgeChunks - 1;
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0) {   // every piece of this candidate is in: chain and fold them
    ArgBest pick[2] = {{0, -1}, {0, -1}};
    for (int w = 0; w < 2; ++w) {
      const bool wmax = w == 0 ? del : !del;
      long long off = (long long)atomicAdd(&W.start_acc[jb * 2 + w], 0ull);   // dd at the window's first index
      for (int c = 0; c < kEdgeChunks; ++c) {
        const size_t a = (size_t)jb * 2 * kEdgeChunks + w * kEdgeChunks + c;
        const int pi = (int)ld_cg(reinterpret_cast<const unsigned int*>(&W.part_i[a]));
        if (pi >= 0) {
          const long long v = (long long)ld_cg(reinterpret_cast<const unsigned long long*>(&W.part_v[a])) + off;
          if (wmax ? v > 0 : v < 0) pick[w] = arg_pick(pick[w], ArgBest{v, pi}, wmax);
        }
        off += (long long)ld_cg(reinterpret_cast<const unsigned long long*>(&W.part_t[a]));
      }
      W.start_acc[jb * 2 + w] = 0;   // ready for the second call
    }
    if (pick[0].i > 0) jobs[jb].start = from + pick[0].i;
    if (pick[1].i > 0) jobs[jb].end = to - nstep + pick[1].i;
    W.done[jb] = 0;
  }
}

// ------------------------------------------------------------------------------------------
// Neighbourhood test.  See CandJob in kernels.h for what the host prepares.
constexpr int kWalkPer = 16;                          // consecutive positions per thread and trip: ONE 16-byte load of the byte array
constexpr int kWalkBlock = kWalkPer * kTestThreads;   // positions examined per trip of a walk

// s_scan doubles as scratch of hist_ranks; the rest is the walk's per-trip exchange, double-buffered by trip parity
struct WalkShared { int s_scan[kMaxWaves]; int cnt[2][kMaxWaves]; int trg[2][kMaxWaves]; int kept[2]; int lastt[2]; };

// Sixteen consecutive values in walk order from `first` (first, first + dir, ...): one or four 16-byte loads where the
// block lies inside the array, clamped single loads at the chromosome's ends (those positions are masked by the caller).
struct __attribute__((packed, aligned(1))) WalkBytes16 { uint32_t w[4]; };
struct __attribute__((packed, aligned(4))) WalkInts4 { int x, y, z, w; };
__device__ inline void walk_load16(const uint8_t* __restrict__ A, int64_t N, long long first, int dir, int* out) {
  const long long lo = dir > 0 ? first : first - 15;
  if (lo >= 0 && lo + 15 <= N - 1) {
    const WalkBytes16 b = *reinterpret_cast<const WalkBytes16*>(A + lo);
#pragma unroll
    for (int j = 0; j < 16; ++j) { const int k = dir > 0 ? j : 15 - j; out[j] = (int)((b.w[k >> 2] >> (8 * (k & 3))) & 0xffu); }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) { long long p = first + (long long)dir * j; p = p < 0 ? 0 : (p > N - 1 ? N - 1 : p); out[j] = (int)A[p]; }
  }
}
__device__ inline void walk_load16(const int32_t* __restrict__ A, int64_t N, long long first, int dir, int* out) {
  const long long lo = dir > 0 ? first : first - 15;
  if (lo >= 0 && lo + 15 <= N - 1) {
    int v[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const WalkInts4 b = *reinterpret_cast<const WalkInts4*>(A + lo + 4 * q); v[4 * q] = b.x; v[4 * q + 1] = b.y; v[4 * q + 2] = b.z; v[4 * q + 3] = b.w; }
#pragma unroll
    for (int j = 0; j < 16; ++j) out[j] = v[dir > 0 ? j : 15 - j];
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) { long long p = first + (long long)dir * j; p = p < 0 ? 0 : (p > N - 1 ? N - 1 : p); out[j] = A[p]; }
  }
}

// One side of the reference gather (rsi.cpp:206-257).  dir = -1: left of the candidate, values land
// in dst[fill], dst[fill-1], ...; dir = +1: right, values land in dst[used], dst[used+1], ...
// Returns the number of values stored; *reach = last position examined.
// A trip examines kWalkBlock positions in walk order (sixteen consecutive ones per thread): which are
// taken (not extreme, not inside the neighbour the walk is about to meet), and where the walk first
// steps into that neighbour (the trigger, rsi.cpp:222-228 / 246-252: everything after it is dropped
// and the walk jumps).  One barrier per trip: the waves exchange their taken-counts and triggers; a
// second one only when there is a trigger, a third when the slots run out.
template <typename TD>
__device__ inline int gather_side(const TD* __restrict__ A, int64_t N, int dir, int pos, int room /* slots left */,
                                  int32_t* __restrict__ dst, int first_slot, const int2* __restrict__ chain, int nchain,
                                  int kind, double too_high, double too_low, WalkShared& W, int* reach, int* chain_used,
                                  int* __restrict__ stage /* LDS, kWalkBlock ints: a trip's taken values in rank order */) {
  constexpr int kNone = 0x7fffffff;
  // A thread's sixteen values have consecutive ranks, so a store of value j by the 64 lanes of a wave would touch 64 cache
  // lines 64 bytes apart -- the walk spent most of its time in those stores.  The values go to LDS first (the rank's low
  // four bits XORed with the lane's, which spreads a wave over the banks) and leave as consecutive dwords per wave.
  auto swz = [](int r) { return r ^ ((r >> 4) & 15); };
  int stored = 0, ci = 0, trip = 0;
  int last = pos;
  const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  // "extreme" for integer values: (double)v > too_high <=> v > floor(too_high), (double)v < too_low <=> v < ceil(too_low)
  // (rsi.cpp:213, 237); the limits are clamped into the int range first, which changes no comparison with an int
  const double th = too_high > 2147483646.0 ? 2147483646.0 : (too_high < -2147483647.0 ? -2147483647.0 : floor(too_high));
  const double tl = too_low > 2147483646.0 ? 2147483646.0 : (too_low < -2147483647.0 ? -2147483647.0 : ceil(too_low));
  const int hi_lim = kind == 0 && too_high == too_high ? (too_high > 2147483646.0 ? 0x7fffffff : (int)th) : 0x7fffffff;   // extreme: v > hi_lim (never for a NaN limit)
  const int lo_lim = kind == 1 && too_low == too_low ? (too_low < -2147483647.0 ? (int)0x80000000 : (int)tl) : (int)0x80000000;   // extreme: v < lo_lim
  // the values of a trip are loaded one trip ahead (the walk continues straight on unless it meets a
  // neighbour, which is rare)
  auto load16 = [&](int from, int* out) { walk_load16(A, N, (long long)from + (long long)dir * (1 + kWalkPer * (int)threadIdx.x), dir, out); };
  int vnext[kWalkPer];
  load16(pos, vnext);
  // the neighbour the walk may meet next: read when it changes, not per trip (the list may sit in mapped host memory)
  int2 cur = ci < nchain ? chain[ci] : make_int2(1, 0);   // empty interval when the chain is used up
  while (room > 0 && (dir < 0 ? pos > 2 : (int64_t)pos < N - 2)) {
    const int par = trip & 1;
    ++trip;
    // positions of this trip in walk order: p_t = pos + dir*(1+t)
    const int avail = dir < 0 ? pos - 2 : (int)(N - 2 - pos);     // how many positions the walk may still visit
    const int cnt = avail < kWalkBlock ? avail : kWalkBlock;
    int v[kWalkPer]; unsigned accm = 0; int nacc = 0; int trig = kNone;
#pragma unroll
    for (int j = 0; j < kWalkPer; ++j) v[j] = vnext[j];
    load16(pos + dir * cnt, vnext);                                // next trip, if the walk goes straight on
    const int t0 = kWalkPer * (int)threadIdx.x;
#pragma unroll
    for (int j = 0; j < kWalkPer; ++j) {
      const int t = t0 + j;
      if (t < cnt) {
        const int p = pos + dir * (1 + t);
        const bool ext = v[j] > hi_lim || v[j] < lo_lim;
        const bool inside = p >= cur.x && p <= cur.y;
        if (!ext && inside && t < trig) trig = t;
        if (!ext && !inside) { accm |= 1u << j; ++nacc; }
      }
    }
    int incl = wave_incl_scan(nacc), wtrig = trig;
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(wtrig, d); wtrig = o < wtrig ? o : wtrig; }
    if (lane == 63) W.cnt[par][wave] = incl;
    if (lane == 0) W.trg[par][wave] = wtrig;
    __syncthreads();
    int base = 0, all = 0, tstar = kNone;
    for (int w = 0; w < nwaves; ++w) { const int c = W.cnt[par][w]; if (w < wave) base += c; all += c; const int g = W.trg[par][w]; tstar = g < tstar ? g : tstar; }
    int rank = base + incl - nacc;             // rank of this thread's first taken value if nothing were dropped
    int total = all;
    if (tstar != kNone) {                       // values taken before the trigger = the trigger position's rank
      if ((int)threadIdx.x == tstar / kWalkPer) {
        int k = rank;
#pragma unroll
        for (int j = 0; j < kWalkPer; ++j) if (((accm >> j) & 1u) && t0 + j < tstar) ++k;
        W.kept[par] = k;
      }
      __syncthreads();
      total = W.kept[par];
    }
    const bool fills = total >= room;
#pragma unroll
    for (int j = 0; j < kWalkPer; ++j) {
      const int t = t0 + j;
      if (((accm >> j) & 1u) && t < tstar) {
        if (rank < room) stage[swz(rank)] = v[j];
        if (fills && rank == room - 1) W.lastt[par] = t;      // the value that fills the last slot
        ++rank;
      }
    }
    __syncthreads();
    {
      const int ncopy = total < room ? total : room;
      for (int e = threadIdx.x; e < ncopy; e += blockDim.x) dst[first_slot + dir * (stored + e)] = stage[swz(e)];
    }   // the next trip writes `stage` only behind its own exchange barrier, which every thread reaches after this loop
    if (fills) {   // the walk stops right after the value that filled the last slot
      last = pos + dir * (1 + W.lastt[par]);
      stored += room; room = 0;
      break;
    }
    stored += total; room -= total;
    if (tstar != kNone) {      // jump over the neighbour
      last = pos + dir * (1 + tstar);
      pos = dir < 0 ? cur.x - 1 : cur.y + 1;
      ++ci;
      cur = ci < nchain ? chain[ci] : make_int2(1, 0);
      load16(pos, vnext);      // the prefetch was for the straight continuation
    } else {
      pos += dir * cnt;
      last = pos;
    }
  }
  __syncthreads();   // the next user of W (or of dst) starts from a quiet block
  *reach = last;
  *chain_used = ci;
  return stored;
}

// Buckets where the cumulated count first reaches n/4, n/2, 3n/4 (partition_stat_tp's walk,
// wufunctions.cpp:398-420), -1 where it never does; every thread gets the result.
__device__ inline void hist_ranks(const unsigned int* hist, unsigned nbk, size_t n, int* s_scan, int* s_q, int* out) {
  const size_t r1 = n / 4, r2 = n / 2, r3 = n * 3 / 4;
  if (threadIdx.x < 3) s_q[threadIdx.x] = -1;
  const unsigned chunk = (nbk + blockDim.x - 1) / blockDim.x;
  const unsigned b0 = threadIdx.x * chunk;
  int local = 0;
  for (unsigned b = b0; b < b0 + chunk && b < nbk; ++b) local += (int)hist[b];
  int total;
  size_t seen = (size_t)block_exscan(local, s_scan, &total);   // syncs: s_q is initialised before any write below
  for (unsigned b = b0; b < b0 + chunk && b < nbk; ++b) {
    const size_t upto = seen + hist[b];
    if (seen < r1 && upto >= r1) s_q[0] = (int)b;
    if (seen < r2 && upto >= r2) s_q[1] = (int)b;
    if (seen < r3 && upto >= r3) s_q[2] = (int)b;
    seen = upto;
  }
  __syncthreads();
  out[0] = s_q[0]; out[1] = s_q[1]; out[2] = s_q[2];
  __syncthreads();
}

template <typename TD>
__global__ __launch_bounds__(kTestThreads) void k_candidate_test(const TD* __restrict__ A, int64_t N,
                                                             const CandJob* __restrict__ jobs, const int2* __restrict__ chains,
                                                             int32_t* __restrict__ iscratch, long long* __restrict__ lscratch,
                                                             double RDmedian, CandOut* __restrict__ outs) {
  extern __shared__ unsigned int s_hist[];   // kCandHistBins counters
  __shared__ WalkShared W;
  __shared__ double s_d[kMaxWaves];
  __shared__ float s_f[kMaxWaves];
  __shared__ long long s_l[kMaxWaves];
  __shared__ long long s_x[2][4 * kMaxWaves];   // prefix rounds: wave totals of four tiles, by round parity
  __shared__ int s_i2[kMaxWaves];
  __shared__ int s_q[3];
  const CandJob J = jobs[blockIdx.x];
  CandOut O;
  O.flags = 0;
  // every piece starts on a 16-byte boundary (the host sizes the scratch the same way, cand_scratch_ints)
  int32_t* left = iscratch + J.iscratch_off;                          // J.top + 1 slots
  int32_t* ref = left + (((J.top + 1 > 0 ? J.top + 1 : 0) + 3) & ~3);  // J.capacity slots
  int32_t* thin = ref + ((J.capacity + 3) & ~3);                      // min(capacity, budget) slots
  long long* P = lscratch + J.lscratch_off;               // capacity + 1
  const double too_high = RDmedian * 3.0, too_low = RDmedian * 0.15;

  // ---- gather: left side into left[top..], then ref = left part ++ right part ----
  int lreach = J.start, rreach = J.end;
  int lcnt = 0, lused = 0, rused = 0;
  if (J.top >= 0)
    lcnt = gather_side(A, N, -1, J.start - J.margin, J.top + 1, left, J.top, chains + J.left_off, J.nleft, J.kind, too_high, too_low, W, &lreach, &lused, reinterpret_cast<int*>(s_hist));
  __syncthreads();
  // the reference closes the gap when the left side ran out of sequence, otherwise used = top + 1 (rsi.cpp:231-236)
  const int used0 = lcnt < J.top + 1 ? lcnt : J.top + 1;
  for (int j0 = threadIdx.x; j0 < used0; j0 += 8 * kTestThreads) {   // eight independent loads per round
    int x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int j = j0 + k * kTestThreads; x[k] = left[J.top + 1 - used0 + (j < used0 ? j : j0)]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int j = j0 + k * kTestThreads; if (j < used0) ref[j] = x[k]; }
  }
  __syncthreads();
  int room = J.capacity - used0;
  {   // `used < 2*chklen*d` with the double right-hand side (rsi.cpp:243)
    const int lim = (int)ceil(J.right_cap);
    if (lim - used0 < room) room = lim - used0;
    if (room < 0) room = 0;
  }
  const int rcnt = gather_side(A, N, +1, J.end + J.margin, room, ref, used0, chains + J.right_off, J.nright, J.kind, too_high, too_low, W, &rreach, &rused, reinterpret_cast<int*>(s_hist));
  __syncthreads();
  // a chain the host cut short was consumed to its end: the walk may have missed a neighbour
  if (((J.cut & 1) && lused >= J.nleft) || ((J.cut & 2) && rused >= J.nright)) O.flags |= 8;
  int nref = used0 + rcnt;
  int nbody = J.end - J.start + 1;
  // ---- thinning to about `budget` points (rsi.cpp:264-282) ----
  const int32_t* R = ref;
  int body_len = nbody;       // source length of the body
  bool thin_body = false;
  int nbody_eff = nbody;
  if (nref + nbody > J.budget) {
    const int total = nref + nbody;
    const int tref = (int)((double)nref / (double)total * (double)J.budget);
    const int tbody = (int)((double)nbody / (double)total * (double)J.budget);
    for (int q0 = threadIdx.x; q0 < tref; q0 += 8 * kTestThreads) {   // eight independent loads per round
      int x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int q = q0 + k * kTestThreads; x[k] = ref[(int)((double)(q < tref ? q : q0) / (double)tref * (double)nref)]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int q = q0 + k * kTestThreads; if (q < tref) thin[q] = x[k]; }
    }
    __syncthreads();
    R = thin; nref = tref; thin_body = true; nbody_eff = tbody;
  }
  auto body_at = [&](int q) -> int {
    return (int)(thin_body ? A[J.start + (int)((double)q / (double)nbody_eff * (double)body_len)] : A[J.start + q]);
  };
  const int width = nbody_eff;
  const int nwin = nref - width;
  O.nref = nref; O.nbody = nbody_eff; O.nwin = nwin; O.left_reach = lreach; O.right_reach = rreach;
  if (nwin <= 0 || width <= 0) { O.flags |= 1; if (threadIdx.x == 0) outs[blockIdx.x] = O; return; }

  // ---- body statistics: integer histogram quantiles (partition_stat_tp with dy = 1), sum, sum of squares ----
  {
    int lo = 0x7fffffff, hi = (int)0x80000000; long long s1 = 0, s2 = 0;
    for (int q0 = threadIdx.x; q0 < width; q0 += 8 * kTestThreads) {   // eight independent loads per round (see k_cand_prefix)
      int x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int q = q0 + k * kTestThreads; x[k] = body_at(q < width ? q : q0); }
#pragma unroll
      for (int k = 0; k < 8; ++k) if (q0 + k * kTestThreads < width) { lo = x[k] < lo ? x[k] : lo; hi = x[k] > hi ? x[k] : hi; s1 += x[k]; s2 += (long long)x[k] * x[k]; }
    }
    lo = block_reduce(lo, [](int a, int b) { return a < b ? a : b; }, s_i2);
    hi = block_reduce(hi, [](int a, int b) { return a > b ? a : b; }, s_i2);
    s1 = block_reduce(s1, [](long long a, long long b) { return a + b; }, s_l);
    s2 = block_reduce(s2, [](long long a, long long b) { return a + b; }, s_l);
    O.body_min = lo; O.body_max = hi; O.body_s1 = (double)s1; O.body_s2 = (double)s2;
    O.body_q[0] = lo; O.body_q[1] = (double)s1 / (double)width; O.body_q[2] = hi;
    if ((double)hi - (double)lo >= 1.0) {
      const unsigned nbk = (unsigned)(hi - lo) + 2;
      if (nbk > kCandHistBins) O.flags |= 2;
      else {
        for (unsigned e = threadIdx.x; e < nbk; e += kTestThreads) s_hist[e] = 0;
        __syncthreads();
        for (int q0 = threadIdx.x; q0 < width; q0 += 8 * kTestThreads) {
          int x[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { const int q = q0 + k * kTestThreads; x[k] = body_at(q < width ? q : q0); }
#pragma unroll
          for (int k = 0; k < 8; ++k) if (q0 + k * kTestThreads < width) atomicAdd(&s_hist[x[k] - lo], 1u);
        }
        __syncthreads();
        int qb[3];
        hist_ranks(s_hist, nbk, (size_t)width, W.s_scan, s_q, qb);
        if (qb[0] >= 0) O.body_q[0] = (double)lo + qb[0] * 1.0;
        if (qb[1] >= 0) O.body_q[1] = (double)lo + qb[1] * 1.0;
        if (qb[2] >= 0) O.body_q[2] = (double)lo + qb[2] * 1.0;
        __syncthreads();
      }
    }
  }
  // ---- running mean of width `width` over the neighbourhood (rsi.cpp:113-124): exact prefix, float means ----
  {   // rounds of four tiles (4096 values each, four consecutive values per thread and tile, one 16-byte load each): the
      // four wave scans of a round share ONE barrier, the exchange buffer alternates with the round's parity
    constexpr int kTile = 4 * kTestThreads, kRound = 4 * kTile;
    long long carry = 0;
    if (threadIdx.x == 0) P[0] = 0;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    auto load_tile = [&](int t0) {
      const int e = t0 + 4 * (int)threadIdx.x;
      int4 v = make_int4(0, 0, 0, 0);
      if (e + 3 < nref) v = *reinterpret_cast<const int4*>(R + e);
      else { if (e < nref) v.x = R[e]; if (e + 1 < nref) v.y = R[e + 1]; if (e + 2 < nref) v.z = R[e + 2]; }
      return v;
    };
    int4 nx0 = load_tile(0), nx1 = load_tile(kTile), nx2 = load_tile(2 * kTile), nx3 = load_tile(3 * kTile);
    int par = 0;
    for (int t0 = 0; t0 < nref; t0 += kRound, par ^= 1) {
      const int4 v0 = nx0, v1 = nx1, v2 = nx2, v3 = nx3;
      if (t0 + kRound < nref) {   // the next round's loads fly during this round's scans
        nx0 = load_tile(t0 + kRound); nx1 = load_tile(t0 + kRound + kTile);
        nx2 = load_tile(t0 + kRound + 2 * kTile); nx3 = load_tile(t0 + kRound + 3 * kTile);
      }
      // named variables only: a runtime-indexed array would live in scratch memory
      auto quad = [](const int4& v) { return (long long)v.x + v.y + v.z + v.w; };
      long long i0 = quad(v0), i1 = quad(v1), i2 = quad(v2), i3 = quad(v3);   // inclusive wave scans of the quad sums
      i0 = wave_incl_scan(i0); i1 = wave_incl_scan(i1); i2 = wave_incl_scan(i2); i3 = wave_incl_scan(i3);   // DPP: no LDS round trips
      long long* X = s_x[par];   // [4][kMaxWaves] wave totals
      if (lane == 63) { X[wave] = i0; X[kMaxWaves + wave] = i1; X[2 * kMaxWaves + wave] = i2; X[3 * kMaxWaves + wave] = i3; }
      __syncthreads();
      long long b0 = 0, b1 = 0, b2 = 0, b3 = 0, T0 = 0, T1 = 0, T2 = 0, T3 = 0;
#pragma unroll 1
      for (int w = 0; w < kMaxWaves; ++w) {
        const long long c0 = X[w], c1 = X[kMaxWaves + w], c2 = X[2 * kMaxWaves + w], c3 = X[3 * kMaxWaves + w];
        if (w < wave) { b0 += c0; b1 += c1; b2 += c2; b3 += c3; }
        T0 += c0; T1 += c1; T2 += c2; T3 += c3;
      }
      auto store_quad = [&](int e, const int4& v, long long before) {   // P[e+1..e+4] = before + running sums of v
        const long long a1 = before + v.x, a2 = a1 + v.y, a3 = a2 + v.z, a4 = a3 + v.w;
        if (e < nref) P[e + 1] = a1;
        if (e + 1 < nref) P[e + 2] = a2;
        if (e + 2 < nref) P[e + 3] = a3;
        if (e + 3 < nref) P[e + 4] = a4;
      };
      const int e = t0 + 4 * (int)threadIdx.x;
      store_quad(e, v0, carry + b0 + i0 - quad(v0));
      store_quad(e + kTile, v1, carry + T0 + b1 + i1 - quad(v1));
      store_quad(e + 2 * kTile, v2, carry + T0 + T1 + b2 + i2 - quad(v2));
      store_quad(e + 3 * kTile, v3, carry + T0 + T1 + T2 + b3 + i3 - quad(v3));
      carry += T0 + T1 + T2 + T3;
    }
    __syncthreads();
  }
  const double dw = (double)width;
  // the means are kept (as floats, over the neighbourhood values, which are not needed any more) for the histogram pass
  float* Wm = reinterpret_cast<float*>(ref);
  {
    float flo = 3.0e38f, fhi = -3.0e38f; double m1 = 0, m2 = 0;
    for (int i = threadIdx.x; i < nwin; i += 4 * kTestThreads) {
      long long hi4[4], lo4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int q = i + k * kTestThreads; const int qq = q < nwin ? q : i; hi4[k] = P[qq + width]; lo4[k] = P[qq]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = i + k * kTestThreads;
        if (q < nwin) {
          const float w = (float)((double)(hi4[k] - lo4[k]) / dw);
          Wm[q] = w;
          flo = w < flo ? w : flo; fhi = w > fhi ? w : fhi; m1 += (double)w; m2 += (double)w * (double)w;
        }
      }
    }
    flo = block_reduce(flo, [](float a, float b) { return a < b ? a : b; }, s_f);
    fhi = block_reduce(fhi, [](float a, float b) { return a > b ? a : b; }, s_f);
    m1 = block_reduce(m1, [](double a, double b) { return a + b; }, s_d);
    m2 = block_reduce(m2, [](double a, double b) { return a + b; }, s_d);
    const double lo = flo, hi = fhi;
    O.ref_s1 = m1; O.ref_s2 = m2;
    O.ref_q[0] = lo; O.ref_q[1] = m1 / (double)nwin; O.ref_q[2] = hi;
    if ((hi - lo) >= 0.01) {
      const size_t nbk = (size_t)((hi - lo) / 0.01 + 2);
      if (nbk > kCandHistBins) O.flags |= 4;
      else {
        for (unsigned e = threadIdx.x; e < nbk; e += kTestThreads) s_hist[e] = 0;
        __syncthreads();
        // neighbouring windows differ by one value in `width`, so their means mostly share a bucket:
        // every thread takes eight consecutive means and merges equal buckets before touching LDS
        for (int i0 = 8 * (int)threadIdx.x; i0 < nwin; i0 += 8 * kTestThreads) {
          float w8[8];
          if (i0 + 7 < nwin) {
            const float4 a = *reinterpret_cast<const float4*>(Wm + i0), b = *reinterpret_cast<const float4*>(Wm + i0 + 4);
            w8[0] = a.x; w8[1] = a.y; w8[2] = a.z; w8[3] = a.w; w8[4] = b.x; w8[5] = b.y; w8[6] = b.z; w8[7] = b.w;
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) w8[k] = Wm[i0 + k < nwin ? i0 + k : i0];
          }
          unsigned pend_b = 0xffffffffu, pend_c = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (i0 + k < nwin) {
              const double idx = ((double)w8[k] - lo) / 0.01 + 0.5;   // wufunctions.cpp:396
              const unsigned bkt = (unsigned)(unsigned long long)idx;
              if (bkt != pend_b) { if (pend_c) atomicAdd(&s_hist[pend_b], pend_c); pend_b = bkt; pend_c = 0; }
              ++pend_c;
            }
          }
          if (pend_c) atomicAdd(&s_hist[pend_b], pend_c);
        }
        __syncthreads();
        int qb[3];
        hist_ranks(s_hist, (unsigned)nbk, (size_t)nwin, W.s_scan, s_q, qb);
        if (qb[0] >= 0) O.ref_q[0] = lo + qb[0] * 0.01;
        if (qb[1] >= 0) O.ref_q[1] = lo + qb[1] * 0.01;
        if (qb[2] >= 0) O.ref_q[2] = lo + qb[2] * 0.01;
        __syncthreads();
      }
    }
  }
  if (threadIdx.x == 0) outs[blockIdx.x] = O;
}

// ------------------------------------------------------------------------------------------
// The neighbourhood test over several workgroups.  Same arithmetic as k_candidate_test (see there for the reference
// lines); only the order in which the two float moments of the window means are summed differs (chunks, then a fixed
// order over the chunks).  The neighbourhood is not materialised: ref(j) = left part ++ right part is read in place.
struct CandGeom { int used0, rcnt, nref_raw, nref, nbody, nbody_eff, width, nwin; bool thin; };

__device__ inline CandGeom cand_geometry(const CandJob& J, const CandMid& M) {
  CandGeom g;
  g.used0 = J.top >= 0 ? (M.lcnt < J.top + 1 ? M.lcnt : J.top + 1) : 0;
  int room = J.capacity - g.used0;
  const int lim = (int)ceil(J.right_cap);                  // `used < 2*chklen*d` (rsi.cpp:243)
  if (lim - g.used0 < room) room = lim - g.used0;
  if (room < 0) room = 0;
  g.rcnt = M.rcnt_max < room ? M.rcnt_max : room;
  g.nref_raw = g.used0 + g.rcnt;
  g.nbody = J.end - J.start + 1;
  g.nref = g.nref_raw; g.nbody_eff = g.nbody; g.thin = false;
  if (g.nref_raw + g.nbody > J.budget) {                    // thinning (rsi.cpp:264-282)
    const int total = g.nref_raw + g.nbody;
    g.nref = (int)((double)g.nref_raw / (double)total * (double)J.budget);
    g.nbody_eff = (int)((double)g.nbody / (double)total * (double)J.budget);
    g.thin = true;
  }
  g.width = g.nbody_eff;
  g.nwin = g.nref - g.width;
  return g;
}
struct CandBufs { int32_t *left, *ref, *thin, *right; long long* P; };
__device__ inline CandBufs cand_buffers(const CandJob& J, int32_t* iscratch, long long* lscratch) {
  CandBufs b;
  b.left = iscratch + J.iscratch_off;
  b.ref = b.left + (((J.top + 1 > 0 ? J.top + 1 : 0) + 3) & ~3);
  b.thin = b.ref + ((J.capacity + 3) & ~3);
  const int nthin = J.capacity < J.budget ? J.capacity : J.budget;
  b.right = b.thin + ((nthin + 3) & ~3);
  b.P = lscratch + J.lscratch_off;
  return b;
}
// element i of the (possibly thinned) neighbourhood
__device__ inline int cand_value(const CandJob& J, const CandGeom& g, const CandBufs& b, int i) {
  const int j = g.thin ? (int)((double)i / (double)g.nref * (double)g.nref_raw) : i;
  return j < g.used0 ? b.left[J.top + 1 - g.used0 + j] : b.right[j - g.used0];
}
__device__ inline int cand_chunk_len(int n) { return (((n + kCandChunks - 1) / kCandChunks) + 3) & ~3; }
constexpr int kCandP32MaxChunk = 16000000;   // 255 x this many bytes still fit 32 bits: chunk-local prefixes of byte depth are kept as uint32

// launch 1: grid (2, njobs) -- the left and the right walk of every test side by side
template <typename TD>
__global__ __launch_bounds__(kTestThreads) void k_cand_gather(const TD* __restrict__ A, int64_t N, const CandJob* __restrict__ jobs,
                                                          const int2* __restrict__ chains, int32_t* __restrict__ iscratch,
                                                          long long* __restrict__ lscratch, double RDmedian, CandMid* __restrict__ mid) {
  __shared__ WalkShared W;
  extern __shared__ int s_stage[];   // kWalkBlock ints: a trip's taken values on their way to coalesced stores
  const CandJob J = jobs[blockIdx.y];
  const CandBufs b = cand_buffers(J, iscratch, lscratch);
  const double too_high = RDmedian * 3.0, too_low = RDmedian * 0.15;
  CandMid& M = mid[blockIdx.y];
  if (blockIdx.x == 0) {
    int lreach = J.start, lused = 0, lcnt = 0;
    if (J.top >= 0)
      lcnt = gather_side(A, N, -1, J.start - J.margin, J.top + 1, b.left, J.top, chains + J.left_off, J.nleft, J.kind, too_high, too_low, W, &lreach, &lused, s_stage);
    if (threadIdx.x == 0) { M.lcnt = lcnt; M.lreach = lreach; M.lused = lused; }
  } else {
    // The right walk does not know yet how many slots the left one leaves.  Where the host vouches that the left walk fills
    // all of its top + 1 slots (J.cut bit 2: more than twice as many positions to the left, known neighbours taken off, as
    // slots -- it would take every other base to be extreme for that to fail, and k_cand_hist checks: flag 16 sends the test
    // to the host path) the right walk takes what is left, about half the capacity; else as many as it could ever get.  The
    // right walk is the launch's critical path: a trip of 16 384 positions at a time, twice as many trips as needed before.
    int room = J.capacity;
    const int lim = (int)ceil(J.right_cap);
    if (lim < room) room = lim;
    if ((J.cut & 4) && J.top >= 0) room -= J.top + 1;
    if (room < 0) room = 0;
    int rreach = J.end, rused = 0;
    const int rcnt = gather_side(A, N, +1, J.end + J.margin, room, b.right, 0, chains + J.right_off, J.nright, J.kind, too_high, too_low, W, &rreach, &rused, s_stage);
    if (threadIdx.x == 0) { M.rcnt_max = rcnt; M.rreach = rreach; M.rused = rused; }
  }
}

// launch 2: grid (kCandChunks + 1, njobs) -- chunk-local exact prefix of the neighbourhood (+ chunk totals); the last
// workgroup of a job computes the candidate's own statistics meanwhile
template <typename TD>
__global__ __launch_bounds__(kTestThreads) void k_cand_prefix(const TD* __restrict__ A, const CandJob* __restrict__ jobs,
                                                          int32_t* __restrict__ iscratch, long long* __restrict__ lscratch,
                                                          CandMid* __restrict__ mid) {
  extern __shared__ unsigned int s_hist[];   // kCandHistBins counters (candidate statistics only)
  __shared__ long long s_l[kMaxWaves];
  __shared__ int s_i2[kMaxWaves];
  __shared__ int s_scan[kMaxWaves];
  __shared__ int s_q[3];
  const CandJob J = jobs[blockIdx.y];
  CandMid& M = mid[blockIdx.y];
  const CandGeom g = cand_geometry(J, M);
  const CandBufs b = cand_buffers(J, iscratch, lscratch);
  if (g.nwin <= 0 || g.width <= 0) return;
  if (blockIdx.x < kCandChunks) {
    const int Lc = cand_chunk_len(g.nref);
    const int cb = blockIdx.x * Lc;
    int ce = cb + Lc; if (ce > g.nref) ce = g.nref;
    long long carry = 0;
    // byte depth: a chunk's local prefix stays below 2^24 (62 500 values of at most 255), so it is kept in 32 bits -- half the bytes
    // this launch writes and the next one reads (the prefixes were most of the split form's memory traffic)
    uint32_t* const P32 = reinterpret_cast<uint32_t*>(b.P);
    const bool p32 = sizeof(TD) == 1 && Lc <= kCandP32MaxChunk;
    if (blockIdx.x == 0 && threadIdx.x == 0) { if (p32) P32[0] = 0u; else b.P[0] = 0; }
    // sixteen consecutive values per thread and round (sixteen independent loads, ONE block scan): with four per round the
    // chunk's 62 000 values took sixteen rounds of barriers
    constexpr int kPer = 16;
    for (int t0 = cb; t0 < ce; t0 += kPer * kTestThreads) {
      const int e = t0 + kPer * (int)threadIdx.x;
      int v[kPer];
      if (!g.thin && e + kPer <= ce && (e + kPer <= g.used0 || e >= g.used0)) {
        // sixteen consecutive values from one side of the neighbourhood: four 16-byte loads (a 4-byte load per value makes the
        // 64 lanes of every load touch 64 cache lines, sixteen times over)
        const int32_t* src = e < g.used0 ? b.left + (J.top + 1 - g.used0 + e) : b.right + (e - g.used0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const WalkInts4 w4 = *reinterpret_cast<const WalkInts4*>(src + 4 * q); v[4 * q] = w4.x; v[4 * q + 1] = w4.y; v[4 * q + 2] = w4.z; v[4 * q + 3] = w4.w; }
      } else {
#pragma unroll
        for (int k = 0; k < kPer; ++k) v[k] = e + k < ce ? cand_value(J, g, b, e + k) : 0;
      }
      long long run = 0;
#pragma unroll
      for (int k = 0; k < kPer; ++k) run += v[k];
      long long total;
      long long at = carry + block_exscan_i64(run, s_l, &total);
#pragma unroll
      for (int k = 0; k < kPer; ++k) { at += v[k]; if (e + k < ce) { if (p32) P32[e + k + 1] = (uint32_t)at; else b.P[e + k + 1] = at; } }
      carry += total;
    }
    if (threadIdx.x == 0) M.totals[blockIdx.x] = carry;
    return;
  }
  // ---- candidate statistics: integer histogram quantiles (partition_stat_tp with dy = 1), sum, sum of squares ----
  const int width = g.width;
  auto body_at = [&](int q) -> int {
    return (int)(g.thin ? A[J.start + (int)((double)q / (double)g.nbody_eff * (double)g.nbody)] : A[J.start + q]);
  };
  int lo = 0x7fffffff, hi = (int)0x80000000; long long s1 = 0, s2 = 0;
  if (sizeof(TD) == 1 && !g.thin) {
    // Byte depth, no thinning (the usual case): ONE pass with 16-byte loads -- sixteen values each, four loads in flight per
    // thread -- into a histogram over the byte values themselves; minimum, maximum, sum and sum of squares follow from its 256
    // counters exactly, and the quantile walk starts at the minimum's counter.  The general form below reads the candidate
    // twice, a value per load: 100 us for a candidate of 200 000 bases, the longest workgroup of the launch by far.
    for (unsigned e = threadIdx.x; e < 260; e += kTestThreads) s_hist[e] = 0;
    __syncthreads();
    const uint8_t* B = reinterpret_cast<const uint8_t*>(A) + J.start;
    for (int q0 = 16 * (int)threadIdx.x; q0 < width; q0 += 4 * 16 * kTestThreads) {
      WalkBytes16 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = q0 + k * 16 * kTestThreads;
        if (q + 16 <= width) v[k] = *reinterpret_cast<const WalkBytes16*>(B + q);
        else {
#pragma unroll
          for (int w = 0; w < 4; ++w) v[k].w[w] = 0;
          for (int j = 0; j < 16 && q + j < width; ++j) v[k].w[j >> 2] |= (uint32_t)B[q + j] << (8 * (j & 3));
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = q0 + k * 16 * kTestThreads;
#pragma unroll
        for (int j = 0; j < 16; ++j) if (q + j < width) atomicAdd(&s_hist[(v[k].w[j >> 2] >> (8 * (j & 3))) & 0xffu], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      const long long c = s_hist[threadIdx.x];
      if (c) { lo = (int)threadIdx.x; hi = (int)threadIdx.x; s1 = c * (long long)threadIdx.x; s2 = c * (long long)threadIdx.x * (long long)threadIdx.x; }
    }
    lo = block_reduce(lo, [](int a, int b2) { return a < b2 ? a : b2; }, s_i2);
    hi = block_reduce(hi, [](int a, int b2) { return a > b2 ? a : b2; }, s_i2);
    s1 = block_reduce(s1, [](long long a, long long b2) { return a + b2; }, s_l);
    s2 = block_reduce(s2, [](long long a, long long b2) { return a + b2; }, s_l);
    double q0 = lo, q1 = (double)s1 / (double)width, q2 = hi;
    unsigned flags = 0;
    if ((double)hi - (double)lo >= 1.0) {
      const unsigned nbk = (unsigned)(hi - lo) + 2;
      int qb[3];
      hist_ranks(s_hist + lo, nbk, (size_t)width, s_scan, s_q, qb);
      if (qb[0] >= 0) q0 = (double)lo + qb[0] * 1.0;
      if (qb[1] >= 0) q1 = (double)lo + qb[1] * 1.0;
      if (qb[2] >= 0) q2 = (double)lo + qb[2] * 1.0;
    }
    if (threadIdx.x == 0) {
      M.body_flags = flags; M.body_min = lo; M.body_max = hi; M.body_s1 = (double)s1; M.body_s2 = (double)s2;
      M.body_q[0] = q0; M.body_q[1] = q1; M.body_q[2] = q2;
    }
    return;
  }
  // eight independent loads per thread and round: one after the other, each waiting for memory, the 200 000 values of a large
  // candidate took 120 us through this one workgroup
  for (int q0 = threadIdx.x; q0 < width; q0 += 8 * kTestThreads) {
    int x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int q = q0 + k * kTestThreads; x[k] = body_at(q < width ? q : q0); }
#pragma unroll
    for (int k = 0; k < 8; ++k) if (q0 + k * kTestThreads < width) { lo = x[k] < lo ? x[k] : lo; hi = x[k] > hi ? x[k] : hi; s1 += x[k]; s2 += (long long)x[k] * x[k]; }
  }
  lo = block_reduce(lo, [](int a, int b2) { return a < b2 ? a : b2; }, s_i2);
  hi = block_reduce(hi, [](int a, int b2) { return a > b2 ? a : b2; }, s_i2);
  s1 = block_reduce(s1, [](long long a, long long b2) { return a + b2; }, s_l);
  s2 = block_reduce(s2, [](long long a, long long b2) { return a + b2; }, s_l);
  double q0 = lo, q1 = (double)s1 / (double)width, q2 = hi;
  unsigned flags = 0;
  if ((double)hi - (double)lo >= 1.0) {
    const unsigned nbk = (unsigned)(hi - lo) + 2;
    if (nbk > kCandHistBins) flags = 2;
    else {
      for (unsigned e = threadIdx.x; e < nbk; e += kTestThreads) s_hist[e] = 0;
      __syncthreads();
      for (int q0 = threadIdx.x; q0 < width; q0 += 8 * kTestThreads) {
        int x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int q = q0 + k * kTestThreads; x[k] = body_at(q < width ? q : q0); }
#pragma unroll
        for (int k = 0; k < 8; ++k) if (q0 + k * kTestThreads < width) atomicAdd(&s_hist[x[k] - lo], 1u);
      }
      __syncthreads();
      int qb[3];
      hist_ranks(s_hist, nbk, (size_t)width, s_scan, s_q, qb);
      if (qb[0] >= 0) q0 = (double)lo + qb[0] * 1.0;
      if (qb[1] >= 0) q1 = (double)lo + qb[1] * 1.0;
      if (qb[2] >= 0) q2 = (double)lo + qb[2] * 1.0;
    }
  }
  if (threadIdx.x == 0) {
    M.body_flags = flags; M.body_min = lo; M.body_max = hi; M.body_s1 = (double)s1; M.body_s2 = (double)s2;
    M.body_q[0] = q0; M.body_q[1] = q1; M.body_q[2] = q2;
  }
}

// launch 3: grid (kCandChunks, njobs) -- float window means of a chunk of windows, their extremes and moments
__global__ __launch_bounds__(kTestThreads) void k_cand_means(const CandJob* __restrict__ jobs, int32_t* __restrict__ iscratch,
                                                         long long* __restrict__ lscratch, CandMid* __restrict__ mid,
                                                         int p32 /* the chunk-local prefixes are 32-bit (byte depth, k_cand_prefix) */) {
  __shared__ long long s_off[kCandChunks];
  __shared__ double s_d[kMaxWaves];
  __shared__ float s_f[kMaxWaves];
  const CandJob J = jobs[blockIdx.y];
  CandMid& M = mid[blockIdx.y];
  const CandGeom g = cand_geometry(J, M);
  const CandBufs b = cand_buffers(J, iscratch, lscratch);
  if (g.nwin <= 0 || g.width <= 0) return;
  if (threadIdx.x == 0) { long long o = 0; for (int c = 0; c < kCandChunks; ++c) { s_off[c] = o; o += M.totals[c]; } }
  __syncthreads();
  const int Lc = cand_chunk_len(g.nref);
  const uint32_t* const P32 = reinterpret_cast<const uint32_t*>(b.P);
  const bool use32 = p32 != 0 && Lc <= kCandP32MaxChunk;
  auto P_at = [&](int x) -> long long { return x == 0 ? 0ll : s_off[(x - 1) / Lc] + (use32 ? (long long)P32[x] : b.P[x]); };
  const int Wc = cand_chunk_len(g.nwin);
  const int wb = blockIdx.x * Wc;
  int we = wb + Wc; if (we > g.nwin) we = g.nwin;
  float* Wm = reinterpret_cast<float*>(b.ref);
  const double dw = (double)g.width;
  float flo = 3.0e38f, fhi = -3.0e38f; double m1 = 0, m2 = 0;
  for (int q = wb + (int)threadIdx.x; q < we; q += kTestThreads) {
    const float w = (float)((double)(P_at(q + g.width) - P_at(q)) / dw);
    Wm[q] = w;
    flo = w < flo ? w : flo; fhi = w > fhi ? w : fhi; m1 += (double)w; m2 += (double)w * (double)w;
  }
  flo = block_reduce(flo, [](float a, float b2) { return a < b2 ? a : b2; }, s_f);
  fhi = block_reduce(fhi, [](float a, float b2) { return a > b2 ? a : b2; }, s_f);
  m1 = block_reduce(m1, [](double a, double b2) { return a + b2; }, s_d);
  m2 = block_reduce(m2, [](double a, double b2) { return a + b2; }, s_d);
  if (threadIdx.x == 0) { M.flo[blockIdx.x] = flo; M.fhi[blockIdx.x] = fhi; M.m1[blockIdx.x] = m1; M.m2[blockIdx.x] = m2; }
}

// launch 4: grid (kCandChunks, njobs) -- 0.01-grid histogram of the means; the last workgroup of a job walks it and
// writes the result.  ghist and the ticket are left zero.
__global__ __launch_bounds__(kTestThreads) void k_cand_hist(const CandJob* __restrict__ jobs, int32_t* __restrict__ iscratch,
                                                        long long* __restrict__ lscratch, CandMid* __restrict__ mid,
                                                        uint32_t* __restrict__ ghist_all, CandOut* __restrict__ outs) {
  extern __shared__ unsigned int s_hist[];   // kCandHistBins counters
  __shared__ int s_scan[kMaxWaves];
  __shared__ int s_q[3];
  __shared__ int s_last;
  const CandJob J = jobs[blockIdx.y];
  CandMid& M = mid[blockIdx.y];
  const CandGeom g = cand_geometry(J, M);
  const CandBufs b = cand_buffers(J, iscratch, lscratch);
  uint32_t* ghist = ghist_all + (size_t)blockIdx.y * kCandHistBins;
  CandOut O;
  O.flags = M.body_flags;
  if (((J.cut & 1) && M.lused >= J.nleft) || ((J.cut & 2) && M.rused >= J.nright)) O.flags |= 8;
  if ((J.cut & 4) && J.top >= 0 && M.lcnt < J.top + 1) O.flags |= 16;   // the left walk ran short after all: the right one gathered too little
  O.nref = g.nref; O.nbody = g.nbody_eff; O.nwin = g.nwin; O.left_reach = M.lreach; O.right_reach = M.rreach;
  if (g.nwin <= 0 || g.width <= 0) {
    O.flags = (O.flags & 24) | 1;
    O.body_min = O.body_max = 0; O.body_s1 = O.body_s2 = O.ref_s1 = O.ref_s2 = 0;
    for (int k = 0; k < 3; ++k) { O.body_q[k] = 0; O.ref_q[k] = 0; }
    if (blockIdx.x == 0 && threadIdx.x == 0) outs[blockIdx.y] = O;
    return;
  }
  O.body_min = M.body_min; O.body_max = M.body_max; O.body_s1 = M.body_s1; O.body_s2 = M.body_s2;
  O.body_q[0] = M.body_q[0]; O.body_q[1] = M.body_q[1]; O.body_q[2] = M.body_q[2];
  float flo = 3.0e38f, fhi = -3.0e38f; double m1 = 0, m2 = 0;
  for (int c = 0; c < kCandChunks; ++c) {   // chunks without windows left their neutral elements
    flo = M.flo[c] < flo ? M.flo[c] : flo; fhi = M.fhi[c] > fhi ? M.fhi[c] : fhi; m1 += M.m1[c]; m2 += M.m2[c];
  }
  const double lo = flo, hi = fhi;
  O.ref_s1 = m1; O.ref_s2 = m2;
  O.ref_q[0] = lo; O.ref_q[1] = m1 / (double)g.nwin; O.ref_q[2] = hi;
  const bool spread = (hi - lo) >= 0.01;
  const size_t nbk = spread ? (size_t)((hi - lo) / 0.01 + 2) : 0;
  if (!spread || nbk > kCandHistBins) {
    if (spread) O.flags |= 4;
    if (blockIdx.x == 0 && threadIdx.x == 0) outs[blockIdx.y] = O;
    return;
  }
  const float* Wm = reinterpret_cast<const float*>(b.ref);
  for (unsigned e = threadIdx.x; e < nbk; e += kTestThreads) s_hist[e] = 0;
  __syncthreads();
  const int Wc = cand_chunk_len(g.nwin);
  const int wb = blockIdx.x * Wc;
  int we = wb + Wc; if (we > g.nwin) we = g.nwin;
  // neighbouring windows mostly share a bucket: eight consecutive means per thread, equal buckets merged before the atomic
  for (int i0 = wb + 8 * (int)threadIdx.x; i0 < we; i0 += 8 * kTestThreads) {
    unsigned pend_b = 0xffffffffu, pend_c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (i0 + k < we) {
        const double idx = ((double)Wm[i0 + k] - lo) / 0.01 + 0.5;   // wufunctions.cpp:396
        const unsigned bkt = (unsigned)(unsigned long long)idx;
        if (bkt != pend_b) { if (pend_c) atomicAdd(&s_hist[pend_b], pend_c); pend_b = bkt; pend_c = 0; }
        ++pend_c;
      }
    }
    if (pend_c) atomicAdd(&s_hist[pend_b], pend_c);
  }
  __syncthreads();
  for (unsigned e = threadIdx.x; e < nbk; e += kTestThreads) { const unsigned c = s_hist[e]; if (c) atomicAdd(&ghist[e], c); }
  sync_drained();   // this workgroup's atomics have completed (no fence: device_util.h)
  if (threadIdx.x == 0) s_last = atomicAdd(&M.done, 1u) == kCandChunks - 1;
  __syncthreads();
  if (!s_last) return;
  for (unsigned e = threadIdx.x; e < nbk; e += kTestThreads) s_hist[e] = ld_cg(&ghist[e]);   // the folded histogram, read past the caches
  __syncthreads();
  int qb[3];
  hist_ranks(s_hist, (unsigned)nbk, (size_t)g.nwin, s_scan, s_q, qb);
  if (qb[0] >= 0) O.ref_q[0] = lo + qb[0] * 0.01;
  if (qb[1] >= 0) O.ref_q[1] = lo + qb[1] * 0.01;
  if (qb[2] >= 0) O.ref_q[2] = lo + qb[2] * 0.01;
  for (unsigned e = threadIdx.x; e < nbk; e += kTestThreads) ghist[e] = 0;
  if (threadIdx.x == 0) { M.done = 0; outs[blockIdx.y] = O; }
}

// Sums of depth over inclusive ranges (mean_tp of mergesegments, rsi.cpp:775-779): exact integers.
template <typename TD>
__global__ __launch_bounds__(kThreads) void k_range_sums(const TD* __restrict__ A, const int2* __restrict__ ranges,
                                                         long long* __restrict__ sums) {
  __shared__ long long s_l[kThreads / 64 + 1];
  const int2 r = ranges[blockIdx.x];
  long long acc = 0;
  for (int p = r.x + (int)threadIdx.x; p <= r.y; p += kThreads) acc += A[p];
  acc = block_reduce(acc, [](long long a, long long b) { return a + b; }, s_l);
  if (threadIdx.x == 0) sums[blockIdx.x] = acc;
}

}  // namespace

// The capped, compacted depth is an int32 array or, behind K4', a byte array (DepthRef): every kernel exists for both.
#define RSI_DEPTH_DISPATCH(D, CALL32, CALL8) do { if ((D).bytes == 1) { const uint8_t* rdc = static_cast<const uint8_t*>((D).p); CALL8; } \
                                                 else { const int32_t* rdc = static_cast<const int32_t*>((D).p); CALL32; } } while (0)

void launch_range_sums(DepthRef d, const void* ranges, int nranges, long long* sums, hipStream_t stream) {
  if (nranges <= 0) return;
  RSI_DEPTH_DISPATCH(d,
    RSI_LAUNCH(k_range_sums<int32_t>, dim3(nranges), dim3(kThreads), 0, stream, rdc, static_cast<const int2*>(ranges), sums),
    RSI_LAUNCH(k_range_sums<uint8_t>, dim3(nranges), dim3(kThreads), 0, stream, rdc, static_cast<const int2*>(ranges), sums));
}

void launch_sharpen_edges(DepthRef d, int64_t ncompact, EdgeJob* jobs, int njobs, void* ws, int ws_jobs, hipStream_t stream) {
  if (njobs <= 0) return;
  RSI_DEPTH_DISPATCH(d,
    RSI_LAUNCH(k_sharpen_edges<int32_t>, dim3(2 * kEdgeChunks, njobs), dim3(kThreads), 0, stream, rdc, ncompact, jobs, ws_jobs, ws),
    RSI_LAUNCH(k_sharpen_edges<uint8_t>, dim3(2 * kEdgeChunks, njobs), dim3(kThreads), 0, stream, rdc, ncompact, jobs, ws_jobs, ws));
}
void launch_candidate_test_split(DepthRef d, int64_t ncompact, const CandJob* jobs, int njobs, const void* chains,
                                 int32_t* iscratch, long long* lscratch, double RDmedian, CandMid* mid, uint32_t* ghist,
                                 CandOut* outs, hipStream_t stream) {
  if (njobs <= 0) return;
  const size_t lds = (size_t)kCandHistBins * 4;
  static_assert(kCandHistBins * 4 >= (unsigned)kWalkBlock * 4, "the walks stage a trip in the histogram's LDS");
  RSI_ALLOW_FULL_LDS(k_cand_gather<int32_t>);
  RSI_ALLOW_FULL_LDS(k_cand_gather<uint8_t>);
  RSI_ALLOW_FULL_LDS(k_cand_prefix<int32_t>);
  RSI_ALLOW_FULL_LDS(k_cand_prefix<uint8_t>);
  RSI_ALLOW_FULL_LDS(k_cand_hist);
  const int2* ch = static_cast<const int2*>(chains);
  RSI_DEPTH_DISPATCH(d,
    RSI_LAUNCH(k_cand_gather<int32_t>, dim3(2, njobs), dim3(kTestThreads), lds, stream, rdc, ncompact, jobs, ch, iscratch, lscratch, RDmedian, mid),
    RSI_LAUNCH(k_cand_gather<uint8_t>, dim3(2, njobs), dim3(kTestThreads), lds, stream, rdc, ncompact, jobs, ch, iscratch, lscratch, RDmedian, mid));
  RSI_DEPTH_DISPATCH(d,
    RSI_LAUNCH(k_cand_prefix<int32_t>, dim3(kCandChunks + 1, njobs), dim3(kTestThreads), lds, stream, rdc, jobs, iscratch, lscratch, mid),
    RSI_LAUNCH(k_cand_prefix<uint8_t>, dim3(kCandChunks + 1, njobs), dim3(kTestThreads), lds, stream, rdc, jobs, iscratch, lscratch, mid));
  RSI_LAUNCH(k_cand_means, dim3(kCandChunks, njobs), dim3(kTestThreads), 0, stream, jobs, iscratch, lscratch, mid, d.bytes == 1 ? 1 : 0);
  RSI_LAUNCH(k_cand_hist, dim3(kCandChunks, njobs), dim3(kTestThreads), lds, stream, jobs, iscratch, lscratch, mid, ghist, outs);
}
size_t sharpen_workspace_bytes(int njobs) { return sharpen_zero_bytes(njobs) + (size_t)njobs * 2 * kEdgeChunks * (8 + 8 + 4); }
size_t sharpen_workspace_zero_bytes(int njobs) { return sharpen_zero_bytes(njobs); }
void launch_candidate_test(DepthRef d, int64_t ncompact, const CandJob* jobs, int njobs, const void* chains,
                           int32_t* iscratch, long long* lscratch, double RDmedian, CandOut* outs, hipStream_t stream) {
  if (njobs <= 0) return;
  const size_t lds = (size_t)kCandHistBins * 4;
  RSI_ALLOW_FULL_LDS(k_candidate_test<int32_t>);
  RSI_ALLOW_FULL_LDS(k_candidate_test<uint8_t>);
  RSI_DEPTH_DISPATCH(d,
    RSI_LAUNCH(k_candidate_test<int32_t>, dim3(njobs), dim3(kTestThreads), lds, stream, rdc, ncompact, jobs, static_cast<const int2*>(chains), iscratch, lscratch, RDmedian, outs),
    RSI_LAUNCH(k_candidate_test<uint8_t>, dim3(njobs), dim3(kTestThreads), lds, stream, rdc, ncompact, jobs, static_cast<const int2*>(chains), iscratch, lscratch, RDmedian, outs));
}
#undef RSI_DEPTH_DISPATCH

// The int32 form of a byte array (rsi_hot_fetch("rd_concat"), the host's page fetches): 16 values per thread.
__global__ __launch_bounds__(256) void k_widen_u8(const uint8_t* __restrict__ src, int64_t n, int32_t* __restrict__ dst) {
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (i0 + 16 <= n) {
    const uint4 b = *reinterpret_cast<const uint4*>(src + i0);
    const uint32_t w[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<int4*>(dst + i0 + 4 * q) = make_int4((int)(w[q] & 0xffu), (int)((w[q] >> 8) & 0xffu), (int)((w[q] >> 16) & 0xffu), (int)(w[q] >> 24));
  } else {
    for (int64_t i = i0; i < n; ++i) dst[i] = (int)src[i];
  }
}
void launch_widen_u8(const uint8_t* src, int64_t n, int32_t* dst, hipStream_t stream) {
  if (n <= 0) return;
  const int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
  RSI_LAUNCH(k_widen_u8, dim3((unsigned int)blocks), dim3(256), 0, stream, src, n, dst);
}

// dst[pos[k]] = val[k]: the depths that did not fit a byte, behind launch_widen_u8 (rsi_hot_run's narrowed upload)
__global__ __launch_bounds__(256) void k_patch_i32(int32_t* __restrict__ dst, const int32_t* __restrict__ pos, const int32_t* __restrict__ val, int64_t cnt) {
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * 256) dst[pos[k]] = val[k];
}
void launch_patch_i32(int32_t* dst, const int32_t* pos, const int32_t* val, int64_t cnt, hipStream_t stream) {
  if (cnt <= 0) return;
  const int64_t blocks = std::min<int64_t>((cnt + 255) / 256, 1024);
  RSI_LAUNCH(k_patch_i32, dim3((unsigned int)blocks), dim3(256), 0, stream, dst, pos, val, cnt);
}

}  // namespace rsik
