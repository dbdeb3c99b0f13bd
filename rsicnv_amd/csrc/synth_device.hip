// synth_device.hip -- GPU twin of rsi_synth_generate_host (include/rsi_synth.h): the same
// counter-based generator (synth_core.h) evaluated per base on gfx950, so that multi-gigabase
// inputs for bench.py are produced directly in HBM.  Input generation only (not timed).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "synth_tables.h"

namespace {

struct DevSpec {
  uint64_t seed_fa, seed_rd;
  int64_t n;
  int model, n_events, n_nruns, n_lower;
  const uint32_t* wave;
  const uint64_t* thr;
  const int32_t* off;
  const rsi_synth_interval* events;
  const rsi_synth_interval* nruns;
  const rsi_synth_interval* lower;
};

__global__ __launch_bounds__(256) void k_synth_fasta(DevSpec s, uint8_t* __restrict__ fasta) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < s.n; i += (int64_t)gridDim.x * 256)
    fasta[i] = rsi_synth_base(s.seed_fa, i, s.wave, s.nruns, s.n_nruns, s.lower, s.n_lower);
}

// each thread owns 64 consecutive bases and slides the 201-base GC window along them
__global__ __launch_bounds__(256) void k_synth_depth(DevSpec s, const uint8_t* __restrict__ fasta,
                                                     int32_t* __restrict__ depth) {
  const int64_t chunks = (s.n + 63) / 64;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * 256) {
    const int64_t a = c * 64, b = a + 64 < s.n ? a + 64 : s.n;
    int g = 0;
    if (s.model == 1) {
      const int64_t w0 = a - 100 < 0 ? 0 : a - 100, w1 = a + 100 >= s.n ? s.n - 1 : a + 100;
      for (int64_t j = w0; j <= w1; ++j) g += rsi_is_gc(fasta[j]);
    }
    for (int64_t i = a; i < b; ++i) {
      int cn = RSI_CN_1X;
      const int e = rsi_find_interval(s.events, s.n_events, i);
      if (e >= 0) cn = s.events[e].code;
      depth[i] = fasta[i] == 'N' ? 0
                                 : rsi_synth_depth(s.seed_rd, i, (s.model == 1 ? g : 0) * RSI_CN_CLASSES + cn, s.thr, s.off);
      if (s.model == 1) {
        if (i - 100 >= 0) g -= rsi_is_gc(fasta[i - 100]);
        if (i + 101 < s.n) g += rsi_is_gc(fasta[i + 101]);
      }
    }
  }
}

template <class T>
hipError_t upload(const T* host, size_t count, T** dev) {
  *dev = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(dev), (count ? count : 1) * sizeof(T));
  if (e == hipSuccess && count) e = hipMemcpy(*dev, host, count * sizeof(T), hipMemcpyHostToDevice);
  if (e != hipSuccess) fprintf(stderr, "rsi_synth_generate_device: upload of %zu x %zu bytes failed: %s\n", count, sizeof(T), hipGetErrorString(e));
  return e;
}

}  // namespace

extern "C" int rsi_synth_generate_device(const rsi_synth_spec* spec, void* d_fasta, void* d_depth, void* stream_v) {
  if (!spec || spec->n <= 0 || !d_fasta || !d_depth) return -1;
  if (spec->model != 0 && spec->model != 1) return -2;
  hipStream_t stream = static_cast<hipStream_t>(stream_v);
  SynthTables T;
  synth_build_tables(*spec, T);
  DevSpec s;
  s.seed_fa = rsi_mix64(spec->seed ^ 0xFA57A000ULL);
  s.seed_rd = rsi_mix64(spec->seed ^ 0xDE97B000ULL);
  s.n = spec->n; s.model = spec->model; s.n_events = spec->n_events; s.n_nruns = spec->n_nruns; s.n_lower = spec->n_lower;
  uint32_t* d_wave = nullptr; uint64_t* d_thr = nullptr; int32_t* d_off = nullptr;
  rsi_synth_interval *d_ev = nullptr, *d_nr = nullptr, *d_lo = nullptr;
  const bool ok =
      upload(T.wave.data(), T.wave.size(), &d_wave) == hipSuccess && upload(T.thr.data(), T.thr.size(), &d_thr) == hipSuccess &&
      upload(T.off.data(), T.off.size(), &d_off) == hipSuccess &&
      upload(reinterpret_cast<const rsi_synth_interval*>(spec->events), (size_t)spec->n_events, &d_ev) == hipSuccess &&
      upload(reinterpret_cast<const rsi_synth_interval*>(spec->nruns), (size_t)spec->n_nruns, &d_nr) == hipSuccess &&
      upload(reinterpret_cast<const rsi_synth_interval*>(spec->lower), (size_t)spec->n_lower, &d_lo) == hipSuccess;
  int rc = ok ? 0 : -3;
  if (ok) {
    s.wave = d_wave; s.thr = d_thr; s.off = d_off; s.events = d_ev; s.nruns = d_nr; s.lower = d_lo;
    hipLaunchKernelGGL(k_synth_fasta, dim3(4096), dim3(256), 0, stream, s, static_cast<uint8_t*>(d_fasta));
    hipLaunchKernelGGL(k_synth_depth, dim3(4096), dim3(256), 0, stream, s, static_cast<const uint8_t*>(d_fasta),
                       static_cast<int32_t*>(d_depth));
    if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) rc = -4;
  }
  (void)hipFree(d_wave); (void)hipFree(d_thr); (void)hipFree(d_off); (void)hipFree(d_ev); (void)hipFree(d_nr); (void)hipFree(d_lo);
  return rc;
}
