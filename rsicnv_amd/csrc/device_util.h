// device_util.h -- small device-side building blocks shared by the kernel files: ranges a kernel clears for the kernels
// that follow it in the stream (instead of memset launches), the "last workgroup done" hand-over that lets a kernel finish
// its own reduction (instead of a fold launch), and copies of small results straight into mapped pinned host memory
// (instead of device -> host blits).  A chromosome's chain of launches shrinks to the kernels that do array-sized work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
inline void fill_add(FillList& f, void* p, size_t bytes, unsigned int value) {   // host side; bytes rounded up to 16
  if (!p || bytes == 0 || f.n >= kFillMax) return;
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
__device__ inline void drain() { __builtin_amdgcn_s_waitcnt(0); }   // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's memory operations are done
// barrier after which every thread of the workgroup may ld_cg what any of its threads wrote with st_cg / atomics before it
__device__ inline void sync_drained() { drain(); __syncthreads(); }
__device__ inline void st_cg(unsigned int* p, unsigned int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_cg(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned int ld_cg(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned long long ld_cg(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int ld_cg(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

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
  for (int e = threadIdx.x; e < width; e += blockDim.x) {
    T s = 0;
    const T* col = slabs + (size_t)g * per_group * width + e;
#pragma unroll 8
    for (int k = 0; k < members; ++k) s += ld_cg(col + (size_t)k * width);
    st_cg(gsum + (size_t)g * width + e, s);
  }
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
  for (int e = threadIdx.x; e < width; e += blockDim.x) {
    T s = 0;
#pragma unroll 8
    for (int k = 0; k < ngroups; ++k) s += ld_cg(gsum + (size_t)k * width + e);
    total[e] = s;
  }
  __syncthreads();
  return true;
}

// Whole-workgroup copy of `bytes` (multiple of 4) from device memory written by other workgroups to mapped host memory.
__device__ inline void export_words(void* host_dst, const void* dev_src, size_t bytes) {
  if (!host_dst) return;
  unsigned int* d = static_cast<unsigned int*>(host_dst);
  const unsigned int* s = static_cast<const unsigned int*>(dev_src);
  for (size_t i = threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = ld_cg(s + i);
}
#endif

}  // namespace rsik
