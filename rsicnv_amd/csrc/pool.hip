// pool.hip -- several chromosomes at once on one GPU (the reference's loop over chromosomes, rsi.cpp:2189-2217, whose
// iterations are independent): `nworkers` host threads, each with its own context (stream + workspace), and the gate
// that lets at most two per-base phases stream at a time.
#include "pipeline_internal.h"
#include <condition_variable>
#include <functional>
#include <thread>

using namespace rsik;
using namespace rsip;

extern "C" {

// ------------------------------------------------------------------------------------------
// Pool: `nworkers` host threads, each with its own context (stream + workspace) on one GPU.
// Chromosomes are independent iterations of the reference's loop (rsi.cpp:2189-2217); the pool
// hands them out longest first.
constexpr int kFewChromosomes = 4;   // up to here a run's candidate tests use the multi-workgroup form

struct rsi_pool {
  int device = 0;
  std::vector<rsi_ctx*> workers;
  GpuGate gate;
  std::string err;
  // Workers 1 .. W-1 are threads that live as long as the pool and sleep between runs (worker 0 is the caller's thread):
  // starting eleven threads per run took 0.3 ms before the last chromosome of the first wave was under way.
  std::vector<std::thread> threads;
  std::mutex jm;
  std::condition_variable jcv, dcv;
  const std::function<void(size_t)>* job = nullptr;
  uint64_t generation = 0;
  int pending = 0;
  bool quit = false;
  void worker_loop(size_t w) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(size_t)>* f;
      {
        std::unique_lock<std::mutex> lk(jm);
        jcv.wait(lk, [&] { return quit || generation != seen; });
        if (quit) return;
        seen = generation;
        f = job;
      }
      (*f)(w);
      {
        std::lock_guard<std::mutex> lk(jm);
        if (--pending == 0) dcv.notify_one();
      }
    }
  }
  void run_on_all(const std::function<void(size_t)>& f) {   // f(w) on every worker, w = 0 on the calling thread; returns when all are done
    {
      std::lock_guard<std::mutex> lk(jm);
      job = &f;
      pending = (int)threads.size();
      ++generation;
    }
    jcv.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(jm);
    dcv.wait(lk, [&] { return pending == 0; });
    job = nullptr;
  }
};

rsi_pool* rsi_pool_create(int device, int nworkers, int* status) {
  if (nworkers < 1) nworkers = 1;
  if (nworkers > 64) nworkers = 64;
  rsi_pool* pool = new rsi_pool();
  pool->device = device;
  for (int w = 0; w < nworkers; ++w) {
    int st = 0;
    rsi_ctx* c = rsi_hot_create(device, &st);
    if (!c) {
      if (status) *status = st;
      for (rsi_ctx* x : pool->workers) rsi_hot_destroy(x);
      delete pool;
      return nullptr;
    }
    c->gate = &pool->gate;
    if (const char* ms = getenv("RSI_HOT_STREAMERS")) pool->gate.max_streamers = std::max(1, atoi(ms));
    const char* iso = getenv("RSI_HOT_ISOLATE_STREAMING");
    c->gate_shared = iso && iso[0] == '1';   // default off: bin-level kernels of other chromosomes overlap the per-base phase
    pool->workers.push_back(c);
  }
  for (size_t w = 1; w < pool->workers.size(); ++w) pool->threads.emplace_back([pool, w] { pool->worker_loop(w); });
  if (status) *status = RSI_OK;
  return pool;
}

void rsi_pool_destroy(rsi_pool* pool) {
  if (!pool) return;
  {
    std::lock_guard<std::mutex> lk(pool->jm);
    pool->quit = true;
  }
  pool->jcv.notify_all();
  for (std::thread& t : pool->threads) t.join();
  for (rsi_ctx* c : pool->workers) rsi_hot_destroy(c);
  delete pool;
}

int rsi_pool_workers(const rsi_pool* pool) { return pool ? (int)pool->workers.size() : 0; }
rsi_ctx* rsi_pool_worker(rsi_pool* pool, int w) { return (pool && w >= 0 && w < (int)pool->workers.size()) ? pool->workers[(size_t)w] : nullptr; }
void rsi_pool_set_timing(rsi_pool* pool, int on) { if (pool) for (rsi_ctx* c : pool->workers) rsi_hot_set_timing(c, on); }
void rsi_pool_set_timing_kernel(rsi_pool* pool, const char* name) { if (pool) for (rsi_ctx* c : pool->workers) rsi_hot_set_timing_kernel(c, name); }
void rsi_pool_set_schedule(rsi_pool* pool, int isolate, int streamers) {
  if (!pool) return;
  std::lock_guard<std::mutex> lk(pool->gate.m);
  for (rsi_ctx* c : pool->workers) c->gate_shared = isolate != 0;
  if (streamers >= 1) pool->gate.max_streamers = streamers;
}
const char* rsi_pool_last_error(const rsi_pool* pool) {
  if (pool) return pool->err.c_str();
  static thread_local std::string copy;   // the global record changes under other threads' feet: hand out a snapshot
  { std::lock_guard<std::mutex> lk(g_err_mu); copy = g_last_error; }
  return copy.c_str();
}

static int pool_run_impl(rsi_pool* pool, const rsi_params* p, int nchrom, const void* const* d_depth, const void* const* d_fasta,
                         const int64_t* n, rsi_result** out, int* status, rsi_batch_times* times, bool host_inputs) {
  if (!pool || !p || nchrom < 0 || (nchrom > 0 && (!d_depth || !d_fasta || !n || !out))) return RSI_ERR_BAD_ARG;
  std::vector<int> order((size_t)nchrom);
  for (int i = 0; i < nchrom; ++i) order[(size_t)i] = i;
  {
    int64_t largest = 0;
    for (int i = 0; i < nchrom; ++i) largest = std::max(largest, n[i]);
    for (rsi_ctx* c : pool->workers) c->reserve_n = std::max(c->reserve_n, largest);
  }
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n[a] > n[b]; });
  pool->gate.few_chromosomes = nchrom <= kFewChromosomes;
  std::atomic<int> next(std::min<int>(nchrom, (int)pool->workers.size()));
  std::vector<int> rcs((size_t)nchrom, RSI_OK);
  std::vector<std::vector<std::pair<const char*, float>>> ktimes(pool->workers.size());
  std::vector<std::vector<std::pair<const char*, double>>> ptimes(pool->workers.size());
  std::vector<std::vector<int64_t>> kbases(pool->workers.size());
  static const bool trace = getenv("RSI_HOT_TRACE") && atoi(getenv("RSI_HOT_TRACE")) != 0;   // per-chromosome timeline on stderr
  const double t_run0 = now_ms();
  std::mutex trace_mu;
  auto work = [&](size_t w) {
    rsi_ctx* ctx = pool->workers[w];
    bool first = true;
    for (;;) {
      // the W longest chromosomes always go to the same workers (rank k -> worker k): a context then
      // meets its biggest workload in the first batch and never has to grow its workspace again
      int k;
      if (first && (int)w < nchrom) { k = (int)w; first = false; }
      else { first = false; k = next.fetch_add(1); }
      if (k >= nchrom) break;
      const int i = order[(size_t)k];
      out[i] = nullptr;
      const double t_a = now_ms();
      // host inputs: the worker's own stream carries its chromosome's transfer (pinned memory: a plain DMA), so the transfers of
      // some chromosomes run beside the kernels of others -- H2D double-buffered against compute across the pool's workers
      rcs[(size_t)i] = host_inputs ? rsi_hot_run(ctx, p, static_cast<const int32_t*>(d_depth[i]), static_cast<const uint8_t*>(d_fasta[i]), n[i], &out[i])
                                   : rsi_hot_run_device(ctx, p, d_depth[i], d_fasta[i], n[i], &out[i]);
      if (trace) {
        std::lock_guard<std::mutex> lk(trace_mu);
        fprintf(stderr, "[trace] worker %zu chrom %d n %lld start %.2f end %.2f :", w, i, (long long)n[i], t_a - t_run0, now_ms() - t_run0);
        for (const auto& ph : ctx->phases) fprintf(stderr, " %s=%.2f", ph.first, ph.second);
        fprintf(stderr, "\n");
      }
      if (times) {
        for (const KernelTime& t : ctx->ktimes) { float ms = 0; (void)hipEventElapsedTime(&ms, t.a, t.b); ktimes[w].push_back({t.name, ms}); kbases[w].push_back(n[i]); }
        for (const auto& ph : ctx->phases) ptimes[w].push_back(ph);
      }
    }
  };
  double t_work = 0;
  const std::function<void(size_t)> job = [&](size_t w) { work(w); if (w == 0) t_work = now_ms() - t_run0; };
  pool->run_on_all(job);
  if (trace) fprintf(stderr, "[trace] pool_run: own work done at %.2f ms, all workers joined at %.2f ms\n", t_work, now_ms() - t_run0);
  int worst = RSI_OK;
  for (int i = 0; i < nchrom; ++i) {
    if (status) status[i] = rcs[(size_t)i];
    if (rcs[(size_t)i] != RSI_OK && worst == RSI_OK) { worst = rcs[(size_t)i]; pool->err = g_last_error; }
  }
  if (times) {   // accumulate into the caller's table (names are static strings)
    for (size_t w = 0; w < pool->workers.size(); ++w) {
      for (size_t e = 0; e < ktimes[w].size(); ++e) {
        int slot = -1;
        for (int q = 0; q < times->nkernels; ++q) if (times->kernel_name[q] == ktimes[w][e].first) { slot = q; break; }
        if (slot < 0 && times->nkernels < RSI_MAX_TIMED) { slot = times->nkernels++; times->kernel_name[slot] = ktimes[w][e].first; times->kernel_ms[slot] = 0; times->kernel_launches[slot] = 0; times->kernel_bases[slot] = 0; }
        if (slot >= 0) { times->kernel_ms[slot] += ktimes[w][e].second; times->kernel_launches[slot] += 1; times->kernel_bases[slot] += kbases[w][e]; }
      }
      for (const auto& ph : ptimes[w]) {
        int slot = -1;
        for (int q = 0; q < times->nphases; ++q) if (times->phase_name[q] == ph.first) { slot = q; break; }
        if (slot < 0 && times->nphases < RSI_MAX_TIMED) { slot = times->nphases++; times->phase_name[slot] = ph.first; times->phase_ms[slot] = 0; }
        if (slot >= 0) times->phase_ms[slot] += ph.second;
      }
    }
  }
  return worst;
}

int rsi_pool_run(rsi_pool* pool, const rsi_params* p, int nchrom, const void* const* d_depth, const void* const* d_fasta,
                 const int64_t* n, rsi_result** out, int* status, rsi_batch_times* times) {
  return pool_run_impl(pool, p, nchrom, d_depth, d_fasta, n, out, status, times, false);
}
int rsi_pool_run_host(rsi_pool* pool, const rsi_params* p, int nchrom, const int32_t* const* depth, const uint8_t* const* fasta,
                      const int64_t* n, rsi_result** out, int* status, rsi_batch_times* times) {
  return pool_run_impl(pool, p, nchrom, reinterpret_cast<const void* const*>(depth), reinterpret_cast<const void* const*>(fasta), n, out, status, times, true);
}

}  // extern "C"
