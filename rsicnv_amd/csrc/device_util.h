// device_util.h -- small device-side building blocks shared by the kernel files: ranges a kernel clears for the kernels
// that follow it in the stream (instead of memset launches), the "last workgroup done" hand-over that lets a kernel finish
// its own reduction (instead of a fold launch), and copies of small results straight into mapped pinned host memory
// (instead of device -> host blits).  A chromosome's chain of launches shrinks to the kernels that do array-sized work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

// The hand-over protocol below (write-through `sc1` stores, `s_waitcnt vmcnt(0)` as "my stores are done", hand-written
// `global_load_dwordx4 ... sc1`) is correct on the cache behaviour of gfx942 / gfx950 only: on gfx10 and later stores count in
// vscnt, not vmcnt, and the asm does not assemble or, worse, races silently.  The library is written for gfx950 alone.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "librsi_hot's fence-free hand-over (device_util.h) is only valid on gfx942 / gfx950: build with --offload-arch=gfx950"
#endif

namespace rsik {

// Up to kFillMax ranges of 16-byte units, each set to a 32-bit pattern by the kernel that carries the list, for the
// benefit of LATER kernels in the same stream (nothing in the carrying kernel may depend on them).
constexpr int kFillMax = 6;
struct FillList {
  void* p[kFillMax];              // 16-byte aligned
  unsigned long long units[kFillMax];   // 16-byte units
  unsigned int value[kFillMax];
  int n;
};
// First failure of this thread's launches since the last wait (kernels.h: RSI_LAUNCH records, the pipeline's next ctx_sync
// collects and fails the run).  Defined here because a FillList that cannot take another range fails the same way.
inline thread_local hipError_t tl_launch_error = hipSuccess;
inline void fill_add(FillList& f, void* p, size_t bytes, unsigned int value) {   // host side; bytes rounded up to 16
  if (!p || bytes == 0) return;
  if (f.n >= kFillMax) {   // a dropped range = stale counters behind it (wrong medians, a hand-over that never completes): never
                           // silent, and never the host application's death either -- the run that carries the list fails
    fprintf(stderr, "librsi_hot: FillList overflow (more than %d ranges): raise kFillMax\n", kFillMax);
    if (tl_launch_error == hipSuccess) tl_launch_error = hipErrorInvalidValue;
    return;
  }
  f.p[f.n] = p; f.units[f.n] = (bytes + 15) / 16; f.value[f.n] = value; ++f.n;
}

#if defined(__HIPCC__)
__device__ inline void fill_ranges(const FillList& f) {
  const unsigned long long t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long step = (unsigned long long)gridDim.x * blockDim.x;
  for (int k = 0; k < f.n; ++k) {
    uint4* p = reinterpret_cast<uint4*>(f.p[k]);
    const unsigned int v = f.value[k];
    const uint4 v4 = make_uint4(v, v, v, v);
    for (unsigned long long i = t0; i < f.units[k]; i += step) p[i] = v4;
  }
}

// Hand-over between workgroups of ONE launch without fences.  An agent-scope fence (__threadfence) writes back the whole
// L2 of the XCD on gfx942 / gfx950 (buffer_wbl2) -- with a streaming kernel's dirty output in it that costs microseconds
// per workgroup, a millisecond per launch.  Instead: everything one workgroup hands to another is written with st_cg
// (agent-scope store, write-through to the coherence point) or with atomics; every wave of the writer waits for its own
// stores and atomics to complete (drain(): s_waitcnt vmcnt(0) -- a workgroup barrier alone does not wait for global
// stores on this target), the workgroup meets at a barrier, and only then does one lane bump the arrival counter; the
// reader uses ld_cg (agent-scope load).  Plain loads of data other workgroups of the same launch wrote with plain stores
// are NOT safe and not used.
// -DRSI_HOT_FENCES=1 (make FENCES=1 -> librsi_hot_fences.so) is the debug form: every drain also runs a full agent-scope
// fence (write-back + invalidate), i.e. the textbook release / acquire on both sides of every hand-over, for A/B validation of
// results against the fence-free build (RSI_HOT_LIB selects the library; tools/ab_bench.py compares two in one process).
#if defined(RSI_HOT_FENCES) && RSI_HOT_FENCES
__device__ inline void drain() { __builtin_amdgcn_s_waitcnt(0); __threadfence(); }
#else
__device__ inline void drain() { __builtin_amdgcn_s_waitcnt(0); }   // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's memory operations are done
#endif
// barrier after which every thread of the workgroup may ld_cg what any of its threads wrote with st_cg / atomics before it
__device__ inline void sync_drained() { drain(); __syncthreads(); }
__device__ inline void st_cg(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_cg(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned int ld_cg(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned long long ld_cg(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int ld_cg(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Inclusive prefix sum over the 64 lanes of a wave without the LDS pipeline (`__shfl_up` compiles to ds_bpermute: a round trip
// through the LDS crossbar per step): four row shifts inside the rows of 16 lanes, then lane 15 of each row into the next row and
// lane 31 into the upper half -- six DPP moves.  Integer sums only (the association differs from the shuffle form).  All 64
// lanes must be active.
__device__ inline int wave_incl_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);    // row_shr:1, zeros shifted in
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);    // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);    // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);    // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
  return x;
}
template <int CTRL, int ROWS, bool BOUND>
__device__ inline long long dpp_move_i64(long long x) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)(unsigned long long)x, CTRL, ROWS, 0xF, BOUND);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned int)((unsigned long long)x >> 32), CTRL, ROWS, 0xF, BOUND);
  return (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo);
}
__device__ inline long long wave_incl_scan(long long x) {
  x += dpp_move_i64<0x111, 0xF, true>(x);
  x += dpp_move_i64<0x112, 0xF, true>(x);
  x += dpp_move_i64<0x114, 0xF, true>(x);
  x += dpp_move_i64<0x118, 0xF, true>(x);
  x += dpp_move_i64<0x142, 0xA, false>(x);
  x += dpp_move_i64<0x143, 0xC, false>(x);
  return x;
}

// The same for doubles (the scan kernels' tile prefixes: sums that are exact in double, so the association does not matter), and
// the value of lane 63 through v_readlane instead of a shuffle.
__device__ inline double wave_incl_scan(double x) {
  x += __longlong_as_double(dpp_move_i64<0x111, 0xF, true>(__double_as_longlong(x)));
  x += __longlong_as_double(dpp_move_i64<0x112, 0xF, true>(__double_as_longlong(x)));
  x += __longlong_as_double(dpp_move_i64<0x114, 0xF, true>(__double_as_longlong(x)));
  x += __longlong_as_double(dpp_move_i64<0x118, 0xF, true>(__double_as_longlong(x)));
  x += __longlong_as_double(dpp_move_i64<0x142, 0xA, false>(__double_as_longlong(x)));
  x += __longlong_as_double(dpp_move_i64<0x143, 0xC, false>(__double_as_longlong(x)));
  return x;
}
// Wave-wide maximum of unsigned values the same way (zeros shifted in are the identity); the result is in lane 63.
__device__ inline unsigned int wave_incl_max(unsigned int x) {
  auto mx = [](unsigned int a, int b) { return (unsigned int)b > a ? (unsigned int)b : a; };
  x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true));
  x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true));
  x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true));
  x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true));
  x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false));
  x = mx(x, __builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false));
  return x;
}
__device__ inline long long wave_lane63(long long x) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)x, 63);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)x >> 32), 63);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ inline int wave_lane63(int x) { return __builtin_amdgcn_readlane(x, 63); }
__device__ inline double wave_lane63(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(unsigned long long)b, 63);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)((unsigned long long)b >> 32), 63);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// True in exactly one workgroup of a 1-D launch: the one that arrives last, after every other workgroup's global writes
// and atomics are visible.  *counter must be zero before the launch and is zero again afterwards.  All threads of every
// workgroup must call it (it synchronises the workgroup).
__device__ inline bool last_block_done(unsigned int* counter) {
  __shared__ unsigned int s_last__;
  drain();
  __syncthreads();   // this workgroup's st_cg stores and atomics have completed
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(counter, 1u);
    const bool last = t == gridDim.x * gridDim.y - 1;
    if (last) atomicExch(counter, 0u);
    s_last__ = last ? 1u : 0u;
  }
  __syncthreads();
  return s_last__ != 0;
}

// Two-level hand-over of per-workgroup partial results.  Every workgroup has written its `slab` (width words, st_cg
// stores) at slabs + blockIdx.x * width.  Workgroups form groups of `per_group` consecutive indices; the last one of a
// group to arrive sums the group's slabs into gsum + group * width, and the last group to arrive sums the group sums
// into total[] (global or LDS) and gets `true`: exactly one workgroup of the launch, after all others are done.  No
// atomics on the data, no spinning: only arrival counters (counters[0 .. ngroups], zero before the launch and again
// after it).  All threads of every workgroup must call it.  Same-address atomics from thousands of workgroups, the
// obvious alternative, serialise on the memory side and cost more than a whole streaming pass.
constexpr int kFoldGroups = 32;   // upper bound on the number of groups
inline int fold_per_group(int nblocks) { return (nblocks + kFoldGroups - 1) / kFoldGroups; }
// fold_slabs_add has no second reading level (the groups' sums are atomic adds): more, smaller groups make its one level shorter
constexpr int kFoldGroupsAdd = 64;
inline int fold_per_group_add(int nblocks) { return (nblocks + kFoldGroupsAdd - 1) / kFoldGroupsAdd; }

// Sixteen bytes from each of eight slabs with agent-scope coherence (the sc1 bit an agent-scope atomic load carries), all
// eight in flight at once.  The slab bases are wave-uniform (scalar registers), the thread's offset is one VGPR.  An atomic
// load is a dword, and sixteen of them per thread and slab, eight in flight, made the fold the longest part of K4's tail.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned long long uniform_address(const void* p) {   // the same in every lane: make the compiler keep it in scalar registers
  const unsigned long long a = (unsigned long long)p;
  return (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a) |
         ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(a >> 32)) << 32);
}
__device__ inline void ld_cg_x8(u32x4 (&v)[8], unsigned int voff, const unsigned long long (&b)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, %9 sc1\n\t"
      "global_load_dwordx4 %1, %8, %10 sc1\n\t"
      "global_load_dwordx4 %2, %8, %11 sc1\n\t"
      "global_load_dwordx4 %3, %8, %12 sc1\n\t"
      "global_load_dwordx4 %4, %8, %13 sc1\n\t"
      "global_load_dwordx4 %5, %8, %14 sc1\n\t"
      "global_load_dwordx4 %6, %8, %15 sc1\n\t"
      "global_load_dwordx4 %7, %8, %16 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(voff), "s"(b[0]), "s"(b[1]), "s"(b[2]), "s"(b[3]), "s"(b[4]), "s"(b[5]), "s"(b[6]), "s"(b[7])
      : "memory");
}
__device__ inline void fold_add(unsigned int (&s)[4], u32x4 v) { s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w; }
__device__ inline void fold_add(unsigned long long (&s)[2], u32x4 v) {
  s[0] += (unsigned long long)v.x | ((unsigned long long)v.y << 32);
  s[1] += (unsigned long long)v.z | ((unsigned long long)v.w << 32);
}
// Sixteen coherent bytes through a buffer descriptor (`buffer_load_dwordx4 ... sc1`: the agent-scope load as above), issued by
// the COMPILER: it keeps count of the loads in flight itself, so a loop over members can have sixteen and more outstanding per
// thread where the inline-asm forms above wait after every eighth.  The descriptor's base is wave-uniform; `soff` must be too.
__device__ inline __amdgpu_buffer_rsrc_t coherent_buffer(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(uniform_address(base)), 0, 0x7fffffff, 0x00020000);
}
__device__ inline u32x4 ld_cg_buf_x4(__amdgpu_buffer_rsrc_t r, unsigned int voff, unsigned int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 16 /* sc1 */);
}
// dst[e] = sum over k < members of src[k * stride + e], e < width; width a multiple of 16 / sizeof(T), src 16-byte aligned,
// members * stride * sizeof(T) < 2^31.  kFlight members' quads in flight per thread (sixteen in fold_slabs_add): the fold is a chain of memory round trips
// (2 - 3 us each, the slabs were written by other XCDs), and with eight in flight K4j's last group spent sixteen of them here.
template <typename T, int MODE /* 0: plain stores, 1: st_cg stores, 2: atomic adds of the non-zero sums */, int kFlight = 8>
__device__ inline void fold_columns(const T* src, size_t stride, int members, int width, T* dst) {
  constexpr int kPer = 16 / (int)sizeof(T);
  const int last = __builtin_amdgcn_readfirstlane(members - 1);
  const __amdgpu_buffer_rsrc_t rs = coherent_buffer(src);
  const unsigned int sbytes = (unsigned int)(stride * sizeof(T));
  for (int q = threadIdx.x; q < width / kPer; q += blockDim.x) {
    T s[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) s[j] = 0;
    for (int k0 = 0; k0 <= last; k0 += kFlight) {
      u32x4 v[kFlight];
#pragma unroll
      for (int j = 0; j < kFlight; ++j) v[j] = ld_cg_buf_x4(rs, (unsigned int)q * 16u, (unsigned int)(k0 + j < last ? k0 + j : last) * sbytes);
#pragma unroll
      for (int j = 0; j < kFlight; ++j) if (k0 + j <= last) fold_add(s, v[j]);
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (MODE == 2) { if (s[j]) atomicAdd(dst + q * kPer + j, s[j]); }
      else if (MODE == 1) st_cg(dst + q * kPer + j, s[j]);
      else dst[q * kPer + j] = s[j];
    }
  }
}

template <typename T>
__device__ inline bool fold_slabs(const T* slabs, T* gsum, T* total, int width, int per_group, unsigned int* counters) {
  __shared__ unsigned int s_flag__;
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
    s_flag__ = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_flag__) return false;
  fold_columns<T, 1>(slabs + (size_t)g * per_group * width, (size_t)width, members, width, gsum + (size_t)g * width);
  drain();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(&counters[0], 1u);
    const bool last = t == (unsigned int)ngroups - 1u;
    if (last) atomicExch(&counters[0], 0u);
    s_flag__ = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_flag__) return false;
  fold_columns<T, 0>(gsum, (size_t)width, ngroups, width, total);
  __syncthreads();
  return true;
}

// The same with ONE level of reading: the last workgroup of a group adds the group's sums to total[] (global, zero before
// the launch) with atomics -- a few dozen per word in all -- instead of leaving them for a second pass by a single
// workgroup, which for a wide slab (K4': 16 KB) is as long again as the first (half a megabyte through one workgroup's
// loads).  True in the workgroup that arrives last, after every group's atomics have completed; read total[] with ld_cg.
template <typename T>
__device__ inline bool fold_slabs_add(const T* slabs, T* total, int width, int per_group, unsigned int* counters) {
  __shared__ unsigned int s_flag2__;
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
    s_flag2__ = last ? 1u : 0u;
  }
  __syncthreads();
  if (!s_flag2__) return false;
  fold_columns<T, 2, 16>(slabs + (size_t)g * per_group * width, (size_t)width, members, width, total);
  drain();
  __syncthreads();   // this group's atomics have completed
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(&counters[0], 1u);
    const bool last = t == (unsigned int)ngroups - 1u;
    if (last) atomicExch(&counters[0], 0u);
    s_flag2__ = last ? 1u : 0u;
  }
  __syncthreads();
  return s_flag2__ != 0;
}

// Whole-workgroup copy of `bytes` (multiple of 4) from device memory written by other workgroups to mapped host memory.
// Sixteen bytes per load and store, four loads in flight per thread, where both ends are 16-byte aligned: K4''s 15 KB
// histogram went out as four thousand dword loads, sixteen dependent rounds per thread.
__device__ inline void export_words(void* host_dst, const void* dev_src, size_t bytes) {
  if (!host_dst) return;
  size_t done = 0;
  if (((reinterpret_cast<uintptr_t>(host_dst) | reinterpret_cast<uintptr_t>(dev_src)) & 15) == 0) {
    const size_t nvec = bytes / 16;
    u32x4* d = static_cast<u32x4*>(host_dst);
    const unsigned long long base = uniform_address(dev_src);
    for (size_t q0 = 0; q0 < nvec; q0 += 4 * (size_t)blockDim.x) {
      u32x4 v[4];
      unsigned int off[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const size_t q = q0 + (size_t)j * blockDim.x + threadIdx.x; off[j] = (unsigned int)((q < nvec ? q : 0) * 16); }
      asm volatile(
          "global_load_dwordx4 %0, %4, %8 sc1\n\t"
          "global_load_dwordx4 %1, %5, %8 sc1\n\t"
          "global_load_dwordx4 %2, %6, %8 sc1\n\t"
          "global_load_dwordx4 %3, %7, %8 sc1\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
          : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(base)
          : "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) { const size_t q = q0 + (size_t)j * blockDim.x + threadIdx.x; if (q < nvec) d[q] = v[j]; }
    }
    done = nvec * 16;
  }
  unsigned int* d = static_cast<unsigned int*>(host_dst);
  const unsigned int* s = static_cast<const unsigned int*>(dev_src);
  for (size_t i = done / 4 + threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = ld_cg(s + i);
}
#endif

}  // namespace rsik
