// pool.hip -- several chromosomes at once on one GPU (the reference's loop over chromosomes, rsi.cpp:2189-2217, whose
// iterations are independent): `nworkers` host threads, each with its own context (stream + workspace), the gate that
// lets at most three per-base phases stream at a time, and a queue of runs (samples) whose chromosomes the workers take in
// submission order.
#include "pipeline_internal.h"
#include "run_queue.h"
#include "retire.h"
#include <memory>
#include <thread>

using namespace rsik;
using namespace rsip;

extern "C" {

// ------------------------------------------------------------------------------------------
// Pool: `nworkers` host threads, each with its own context (stream + workspace) on one GPU.
// Chromosomes are independent iterations of the reference's loop (rsi.cpp:2189-2217); the pool
// hands them out longest first.
constexpr int kFewChromosomes = 4;   // up to here a run's candidate tests use the multi-workgroup form

// One call's worth of chromosomes.  Runs queue up in submission order; a worker takes the next unclaimed chromosome of the
// oldest run that has one, so a worker that finds nothing left in one run starts on the next: consecutive samples overlap, and
// the last chromosomes of a run do not leave eleven workers idle.
struct PoolRun : rsip::RunBase {   // nitems = chromosomes
  rsi_params params;
  std::vector<const void*> depth, fasta;
  std::vector<int64_t> n;
  rsi_result** out = nullptr;
  int* status = nullptr;
  rsi_batch_times* times = nullptr;
  bool host_inputs = false;
  std::vector<int> order, rcs;
  int worst = RSI_OK;
  double t0 = 0;
  std::vector<std::vector<std::pair<const char*, float>>> ktimes;
  std::vector<std::vector<std::pair<const char*, double>>> ptimes;
  std::vector<std::vector<int64_t>> kbases;
  std::vector<std::string> errs;   // per chromosome: the failing call's own message (written by the worker that ran it)
};

struct rsi_pool {
  int device = 0;
  std::vector<rsi_ctx*> workers;
  GpuGate gate;
  std::string err;
  // Workers 1 .. W-1 are threads that live as long as the pool and sleep while no run is queued; worker 0's context is used by
  // whichever caller is waiting for a run (rsi_pool_run, rsi_pool_wait): starting eleven threads per run took 0.3 ms before
  // the last chromosome of the first wave was under way.
  std::vector<std::thread> threads;
  rsip::RunQueue<PoolRun> queue;                               // run_queue.h: runs in submission order, items claimed oldest run first
  std::mutex trace_mu;
  mutable std::mutex err_mu;                                   // `err`
  rsip::RetirePolicy retire;                                   // retire.h: who stops claiming when contexts are poisoned

  // A worker whose context has been poisoned (a wait gave up with kernels still queued) fails every run at once: it stops
  // claiming -- unless nobody with a thread of its own would be left to drain the queue (retire.h).  A retirement is the pool's
  // error text until a run's own failure replaces it.
  bool retire_if_poisoned(size_t w) {
    return retire.should_stop(w, workers[w]->poisoned, [this](size_t who) {
      std::lock_guard<std::mutex> lk(err_mu);
      err = "pool worker " + std::to_string(who) + " retired: its context is unusable (an earlier wait gave up with kernels still queued)";
      set_global_error(err);
    });
  }

  void process(size_t w, PoolRun& R, int k) {
    static const bool trace = getenv("RSI_HOT_TRACE") && atoi(getenv("RSI_HOT_TRACE")) != 0;   // per-chromosome timeline on stderr
    rsi_ctx* ctx = workers[w];
    const int i = R.order[(size_t)k];
    R.out[i] = nullptr;
    const double t_a = now_ms();
    struct InFlight { GpuGate& g; InFlight(GpuGate& g_) : g(g_) { g.in_flight.fetch_add(1); } ~InFlight() { g.in_flight.fetch_sub(1); } } in_flight(gate);
    // host inputs: the worker's own stream carries its chromosome's transfer (pinned memory: a plain DMA), so the transfers of
    // some chromosomes run beside the kernels of others -- H2D double-buffered against compute across the pool's workers
    R.rcs[(size_t)i] = R.host_inputs ? rsi_hot_run(ctx, &R.params, static_cast<const int32_t*>(R.depth[(size_t)i]), static_cast<const uint8_t*>(R.fasta[(size_t)i]), R.n[(size_t)i], &R.out[i])
                                     : rsi_hot_run_device(ctx, &R.params, R.depth[(size_t)i], R.fasta[(size_t)i], R.n[(size_t)i], &R.out[i]);
    if (R.rcs[(size_t)i] != RSI_OK) R.errs[(size_t)i] = ctx->err;   // this call's message, not whatever the process saw last
    if (trace) {
      std::lock_guard<std::mutex> lk(trace_mu);
      fprintf(stderr, "[trace] run %llu worker %zu chrom %d n %lld start %.2f end %.2f :", (unsigned long long)R.id, w, i, (long long)R.n[(size_t)i], t_a - R.t0, now_ms() - R.t0);
      for (const auto& ph : ctx->phases) fprintf(stderr, " %s=%.2f", ph.first, ph.second);
      fprintf(stderr, "\n");
    }
    if (R.times) {   // this worker's own rows: nobody else touches them
      for (const KernelTime& t : ctx->ktimes) { float ms = 0; (void)hipEventElapsedTime(&ms, t.a, t.b); R.ktimes[w].push_back({t.name, ms}); R.kbases[w].push_back(R.n[(size_t)i]); }
      for (const auto& ph : ctx->phases) R.ptimes[w].push_back(ph);
    }
  }
  void finish(PoolRun& R) {   // under the queue's mutex, once per run: statuses, error text, the caller's timing table
    static const bool trace = getenv("RSI_HOT_TRACE") && atoi(getenv("RSI_HOT_TRACE")) != 0;
    if (trace) fprintf(stderr, "[trace] run %llu: all chromosomes done at %.2f ms\n", (unsigned long long)R.id, now_ms() - R.t0);
    for (int i = 0; i < R.nitems; ++i) {
      if (R.status) R.status[i] = R.rcs[(size_t)i];
      if (R.rcs[(size_t)i] != RSI_OK && R.worst == RSI_OK) { R.worst = R.rcs[(size_t)i]; std::lock_guard<std::mutex> lk(err_mu); err = R.errs[(size_t)i]; }
    }
    if (rsi_batch_times* times = R.times) {   // accumulate into the caller's table (names are static strings)
      for (size_t w = 0; w < workers.size(); ++w) {
        for (size_t e = 0; e < R.ktimes[w].size(); ++e) {
          int slot = -1;
          for (int q = 0; q < times->nkernels; ++q) if (times->kernel_name[q] == R.ktimes[w][e].first) { slot = q; break; }
          if (slot < 0 && times->nkernels < RSI_MAX_TIMED) { slot = times->nkernels++; times->kernel_name[slot] = R.ktimes[w][e].first; times->kernel_ms[slot] = 0; times->kernel_launches[slot] = 0; times->kernel_bases[slot] = 0; }
          if (slot >= 0) { times->kernel_ms[slot] += R.ktimes[w][e].second; times->kernel_launches[slot] += 1; times->kernel_bases[slot] += R.kbases[w][e]; }
        }
        for (const auto& ph : R.ptimes[w]) {
          int slot = -1;
          for (int q = 0; q < times->nphases; ++q) if (times->phase_name[q] == ph.first) { slot = q; break; }
          if (slot < 0 && times->nphases < RSI_MAX_TIMED) { slot = times->nphases++; times->phase_name[slot] = ph.first; times->phase_ms[slot] = 0; }
          if (slot >= 0) times->phase_ms[slot] += ph.second;
        }
      }
    }
  }
  void worker_loop(size_t w) {
    for (;;) {
      std::shared_ptr<PoolRun> r;
      int k = 0;
      if (!queue.next(r, k)) return;
      process(w, *r, k);
      queue.item_done(r, [this](PoolRun& R) { finish(R); });
      if (retire_if_poisoned(w)) return;
    }
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
  pool->retire.reset(pool->workers.size());
  for (size_t w = 1; w < pool->workers.size(); ++w) pool->threads.emplace_back([pool, w] { pool->worker_loop(w); });
  if (status) *status = RSI_OK;
  return pool;
}

void rsi_pool_destroy(rsi_pool* pool) {
  if (!pool) return;
  pool->queue.shutdown();
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
  static thread_local std::string copy;   // the records change under other threads' feet: hand out a snapshot
  if (pool) { std::lock_guard<std::mutex> lk(pool->err_mu); copy = pool->err; }
  else { std::lock_guard<std::mutex> lk(g_err_mu); copy = g_last_error; }
  return copy.c_str();
}

static uint64_t pool_submit_impl(rsi_pool* pool, const rsi_params* p, int nchrom, const void* const* d_depth, const void* const* d_fasta,
                                 const int64_t* n, rsi_result** out, int* status, rsi_batch_times* times, bool host_inputs, bool caller_first = false) {
  if (!pool || !p || nchrom < 0 || (nchrom > 0 && (!d_depth || !d_fasta || !n || !out))) return 0;
  auto R = std::make_shared<PoolRun>();
  R->params = *p;
  R->nitems = nchrom;
  R->depth.assign(d_depth, d_depth + nchrom);
  R->fasta.assign(d_fasta, d_fasta + nchrom);
  R->n.assign(n, n + nchrom);
  R->out = out; R->status = status; R->times = times; R->host_inputs = host_inputs;
  R->order.resize((size_t)nchrom);
  for (int i = 0; i < nchrom; ++i) R->order[(size_t)i] = i;
  std::stable_sort(R->order.begin(), R->order.end(), [&](int a, int b) { return n[a] > n[b]; });   // longest first
  R->rcs.assign((size_t)nchrom, RSI_OK);
  R->errs.resize((size_t)nchrom);
  R->ktimes.resize(pool->workers.size()); R->ptimes.resize(pool->workers.size()); R->kbases.resize(pool->workers.size());
  R->t0 = now_ms();
  int64_t largest = 0;
  for (int i = 0; i < nchrom; ++i) largest = std::max(largest, n[i]);
  return pool->queue.submit(R, [pool](PoolRun& r) { pool->finish(r); }, [pool, largest, nchrom](PoolRun&) {
    // a context sizes its workspace for the largest chromosome the pool has seen (read when a chromosome starts)
    for (rsi_ctx* c : pool->workers) c->reserve_n = std::max(c->reserve_n.load(), largest);
    pool->gate.few_chromosomes = nchrom <= kFewChromosomes && pool->queue.empty_locked();
  }, caller_first);
}

static int pool_wait_impl(rsi_pool* pool, uint64_t ticket) {
  if (!pool || !ticket) return RSI_ERR_BAD_ARG;
  std::shared_ptr<PoolRun> mine = pool->queue.find(ticket);
  if (!mine) return RSI_ERR_BAD_ARG;
  // the caller is worker 0 while it waits (one such caller at a time; run_queue.h: wait_helping): chromosomes of its own
  // run and of the runs ahead of it, until its run is finished
  pool->queue.wait_helping(mine, [pool](PoolRun& R, int k) { pool->process(0, R, k); }, [pool](PoolRun& R) { pool->finish(R); },
                           [pool] { return !(pool->workers[0]->poisoned && pool->retire_if_poisoned(0)); });
  return mine->worst;
}

int rsi_pool_run(rsi_pool* pool, const rsi_params* p, int nchrom, const void* const* d_depth, const void* const* d_fasta,
                 const int64_t* n, rsi_result** out, int* status, rsi_batch_times* times) {
  const uint64_t t = pool_submit_impl(pool, p, nchrom, d_depth, d_fasta, n, out, status, times, false, true);   // (the caller waits right away)
  return t ? pool_wait_impl(pool, t) : RSI_ERR_BAD_ARG;
}
int rsi_pool_run_host(rsi_pool* pool, const rsi_params* p, int nchrom, const int32_t* const* depth, const uint8_t* const* fasta,
                      const int64_t* n, rsi_result** out, int* status, rsi_batch_times* times) {
  const uint64_t t = pool_submit_impl(pool, p, nchrom, reinterpret_cast<const void* const*>(depth), reinterpret_cast<const void* const*>(fasta), n, out, status, times, true, true);
  return t ? pool_wait_impl(pool, t) : RSI_ERR_BAD_ARG;
}
uint64_t rsi_pool_submit(rsi_pool* pool, const rsi_params* p, int nchrom, const void* const* d_depth, const void* const* d_fasta,
                         const int64_t* n, rsi_result** out, int* status, rsi_batch_times* times) {
  return pool_submit_impl(pool, p, nchrom, d_depth, d_fasta, n, out, status, times, false);
}
int rsi_pool_wait(rsi_pool* pool, uint64_t ticket) { return pool_wait_impl(pool, ticket); }

}  // extern "C"
