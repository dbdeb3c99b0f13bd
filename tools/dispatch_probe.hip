// dispatch_probe.hip -- what a small kernel costs when it is queued behind another one in its stream while the chip is busy.
// S host threads, one stream each, launch N tiny kernels back to back (the chain kernels of a chromosome); beside them a
// background of a chosen kind keeps running on its own streams:
//   none   nothing else
//   spin   persistent ALU workgroups (no memory traffic): W workgroups of 256 threads per CU, ~R VGPRs each
//   stream persistent streaming reads of a large buffer (HBM-bound), W workgroups per CU
//   write  persistent streaming WRITES of a large buffer (plain stores: the write-back L2 fills with dirty lines)
//   burst  R threads launching big kernels back to back that cannot share a CU (W KB of LDS each)
// Prints the mean time per tiny kernel and stream.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/dispatch_probe tools/dispatch_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_tiny(unsigned int* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }

// a small streaming kernel: `wgs` workgroups read n floats and reduce them (a chain kernel with real work)
__global__ void k_small_reduce(const float* __restrict__ x, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += x[i];
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

template <int R>
__global__ __launch_bounds__(256) void k_spin(const volatile int* stop, float* sink, unsigned long long max_cycles) {
  float r[R];
#pragma unroll
  for (int i = 0; i < R; ++i) r[i] = (float)(threadIdx.x + i);
  const unsigned long long t0 = wall_clock64();
  unsigned int iters = 0;
  while (!*stop) {
    ++iters;
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
      for (int i = 0; i < R; ++i) r[i] = r[i] * 1.0001f + r[(i + 1) % R];
    if (wall_clock64() - t0 > max_cycles) break;   // an exit every wave reaches
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) s += r[i];
  if (s == 12345.678f) *sink = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { sink[8] = (float)iters; sink[9] = (float)((wall_clock64() - t0) / 100); }   // loop trips, microseconds resident
}

__global__ __launch_bounds__(256) void k_stream_bg(const uint4* __restrict__ src, size_t n16, const volatile int* stop, unsigned int* sink,
                                                    unsigned long long max_cycles) {
  unsigned int acc = 0, iters = 0;
  const unsigned long long t0 = wall_clock64();
  while (!*stop) {
    ++iters;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
      const uint4 v = src[i];
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (wall_clock64() - t0 > max_cycles) break;
  }
  if (acc == 0x12345u) *sink = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) { reinterpret_cast<float*>(sink)[8] = (float)iters; reinterpret_cast<float*>(sink)[9] = (float)((wall_clock64() - t0) / 100); }
}

__global__ __launch_bounds__(256) void k_write_bg(uint4* __restrict__ dst, size_t n16, const volatile int* stop, unsigned int* sink,
                                                   unsigned long long max_cycles) {
  unsigned int iters = 0;
  const unsigned long long t0 = wall_clock64();
  while (!*stop) {
    ++iters;
    const uint4 v = make_uint4(iters, iters + 1, iters + 2, iters + 3);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;   // plain stores: dirty lines in the write-back L2
    if (wall_clock64() - t0 > max_cycles) break;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { reinterpret_cast<float*>(sink)[8] = (float)iters; reinterpret_cast<float*>(sink)[9] = (float)((wall_clock64() - t0) / 100); }
}

// one of a series of big kernels: every workgroup holds `lds` bytes of LDS and 768 threads for `ticks` of the 100 MHz clock
__global__ __launch_bounds__(768) void k_burst(unsigned long long ticks, float* sink) {
  extern __shared__ float s_buf[];
  s_buf[threadIdx.x] = (float)threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  float a = (float)threadIdx.x;
  while (wall_clock64() - t0 < ticks) { for (int i = 0; i < 64; ++i) a = a * 1.0001f + 1.0f; }
  if (a == 12345.678f) *sink = a + s_buf[0];
}

int main(int argc, char** argv) {
  const char* kind = argc > 1 ? argv[1] : "none";
  const int W = argc > 2 ? atoi(argv[2]) : 4;        // background workgroups per CU
  const int S = argc > 3 ? atoi(argv[3]) : 16;       // foreground streams
  const int N = argc > 4 ? atoi(argv[4]) : 300;      // tiny kernels per stream
  const int R = argc > 5 ? atoi(argv[5]) : 96;       // spin: registers per thread (32, 96)
  const int fg_wgs = argc > 6 ? atoi(argv[6]) : 0;   // 0: the 1-workgroup tiny kernel; > 0: a reduction over 4 MB with that many workgroups
  CHK(hipSetDevice(0));
  int* d_stop; float* d_sink; unsigned int* d_cnt; uint4* d_big = nullptr; float* d_x;
  CHK(hipMalloc(&d_stop, 4)); CHK(hipMalloc(&d_sink, 64)); CHK(hipMalloc(&d_cnt, 4 * 64)); CHK(hipMalloc(&d_x, 4 << 20));
  CHK(hipMemset(d_stop, 0, 4)); CHK(hipMemset(d_cnt, 0, 4 * 64)); CHK(hipMemset(d_x, 0, 4 << 20));
  const size_t big = (size_t)4 << 30;
  hipStream_t bg[3];
  for (auto& s : bg) CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const unsigned long long max_cycles = 100000000ull * 4;   // 4 s at 100 MHz: an exit every wave reaches even if the stop flag never arrives
  if (!strcmp(kind, "spin")) {
    if (R <= 32) hipLaunchKernelGGL(k_spin<24>, dim3(256 * W), dim3(256), 0, bg[0], d_stop, d_sink, max_cycles);
    else hipLaunchKernelGGL(k_spin<96>, dim3(256 * W), dim3(256), 0, bg[0], d_stop, d_sink, max_cycles);
  } else if (!strcmp(kind, "write")) {
    CHK(hipMalloc(&d_big, big)); CHK(hipMemset(d_big, 1, big));
    hipLaunchKernelGGL(k_write_bg, dim3(256 * W), dim3(256), 0, bg[0], d_big, big / 16, d_stop, reinterpret_cast<unsigned int*>(d_sink), max_cycles);
  } else if (!strcmp(kind, "stream")) {
    CHK(hipMalloc(&d_big, big)); CHK(hipMemset(d_big, 1, big));
    hipLaunchKernelGGL(k_stream_bg, dim3(256 * W), dim3(256), 0, bg[0], d_big, big / 16, d_stop, reinterpret_cast<unsigned int*>(d_sink), max_cycles);
  }
  // "burst": B host threads (R = how many) launch big kernels back to back, W = KB of LDS per workgroup, 256 workgroups of 768
  // threads each resident for 150 us: with more LDS than half a CU two of them cannot share a CU, and a launch whose
  // workgroups cannot all be placed waits in the dispatcher
  std::atomic<int> bursts_on(1);
  std::atomic<long> burst_count(0);
  std::vector<std::thread> bth;
  if (!strcmp(kind, "burst")) {
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_burst), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int b = 0; b < R; ++b) bth.emplace_back([&, b] {
      CHK(hipSetDevice(0));
      hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      while (bursts_on.load()) {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k_burst, dim3(256), dim3(768), (size_t)W * 1024, st, 15000ull, d_sink + 2);
        CHK(hipStreamSynchronize(st));
        burst_count += 4;
      }
      CHK(hipStreamDestroy(st));
    });
  }
  CHK(hipGetLastError());
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  std::vector<double> us((size_t)S, 0.0);
  std::vector<std::thread> th;
  std::atomic<int> ready(0);
  for (int s = 0; s < S; ++s) th.emplace_back([&, s] {
    CHK(hipSetDevice(0));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(256), 0, st, d_cnt + s);
    CHK(hipStreamSynchronize(st));
    ++ready; while (ready.load() < S) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) {
      if (fg_wgs > 0) hipLaunchKernelGGL(k_small_reduce, dim3(fg_wgs), dim3(256), 0, st, d_x, (size_t)(1 << 20), d_sink + 1 + (s & 7));
      else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(256), 0, st, d_cnt + s);
    }
    CHK(hipStreamSynchronize(st));
    us[(size_t)s] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
    CHK(hipStreamDestroy(st));
  });
  for (auto& t : th) t.join();
  bursts_on = 0;
  for (auto& t : bth) t.join();
  if (!strcmp(kind, "burst")) printf("    %ld big kernels (%d KB of LDS per workgroup, 150 us each) were launched from %d threads meanwhile\n", burst_count.load(), W, R);
  int one = 1;
  hipStream_t ctl; CHK(hipStreamCreateWithFlags(&ctl, hipStreamNonBlocking));
  CHK(hipMemcpyAsync(d_stop, &one, 4, hipMemcpyHostToDevice, ctl));
  CHK(hipStreamSynchronize(ctl));
  CHK(hipDeviceSynchronize());
  float bgstat[2] = {0, 0};
  CHK(hipMemcpy(bgstat, d_sink + 8, 8, hipMemcpyDeviceToHost));
  double mean = 0, worst = 0;
  for (double v : us) { mean += v / S; worst = v > worst ? v : worst; }
  printf("background %-6s W=%d R=%d | %2d streams x %d %s: %.1f us per kernel and stream (slowest stream %.1f)\n", kind, W, R, S, N,
         fg_wgs > 0 ? "reductions" : "tiny kernels", mean, worst);
  if (strcmp(kind, "none") && strcmp(kind, "burst")) printf("    background workgroup 0: %.0f loop trips in %.0f us%s\n", bgstat[0], bgstat[1],
                                   (!strcmp(kind, "stream") || !strcmp(kind, "write")) ? " (one trip = the 4 GB buffer once)" : "");
  return 0;
}
