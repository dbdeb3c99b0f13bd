// tests/sanitize/retire_tsan.cpp -- the pool's retirement policy (rsicnv_amd/csrc/retire.h) with the pool's queue of runs
// (run_queue.h) and fake "poisoned" flags, under ThreadSanitizer.  ADVICE r4: when every threaded worker was poisoned each saw
// "healthy > 0" (the caller's seat was counted) and exited; runs queued with rsi_pool_submit then made no progress until some
// thread called rsi_pool_wait.  The cases: all but one worker poisoned, then submit WITHOUT wait -- the queue must drain; all
// workers poisoned -- the last claiming thread goes on (failing what it claims), the queue drains, a later wait returns; a pool
// of one (the seat only) -- its occupant goes on.  The worker loop is pool.hip's.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>
#include "../../rsicnv_amd/csrc/retire.h"
#include "../../rsicnv_amd/csrc/run_queue.h"

using namespace rsip;

struct Run : RunBase {
  std::vector<std::atomic<int>> hits;
  std::vector<int> by;   // worker that took the item
  std::atomic<int> finished{0};
  explicit Run(int n) : hits(n), by(n, -1) { nitems = n; for (auto& h : hits) h = 0; }
};

struct FakePool {
  RunQueue<Run> queue;
  RetirePolicy retire;
  std::vector<std::atomic<bool>> poisoned;
  std::vector<std::thread> threads;
  std::atomic<int> retire_msgs{0};
  explicit FakePool(int nworkers) : poisoned(nworkers) {
    for (auto& p : poisoned) p = false;
    retire.reset((size_t)nworkers);
    for (int w = 1; w < nworkers; ++w) threads.emplace_back([this, w] { loop((size_t)w); });
  }
  bool stop(size_t w) { return retire.should_stop(w, poisoned[w].load(), [this](size_t) { retire_msgs.fetch_add(1); }); }
  void process(size_t w, Run& R, int k) {
    R.hits[(size_t)k].fetch_add(1);
    R.by[(size_t)k] = (int)w;
    std::this_thread::sleep_for(std::chrono::microseconds(poisoned[w].load() ? 5 : 50));   // a poisoned context fails at once
  }
  void loop(size_t w) {
    for (;;) {
      std::shared_ptr<Run> r;
      int k = 0;
      if (!queue.next(r, k)) return;
      process(w, *r, k);
      queue.item_done(r, [](Run& R) { R.finished.fetch_add(1); });
      if (stop(w)) return;
    }
  }
  uint64_t submit(const std::shared_ptr<Run>& r) { return queue.submit(r, [](Run& R) { R.finished.fetch_add(1); }, [](Run&) {}); }
  void wait(const std::shared_ptr<Run>& r) {
    queue.wait_helping(r, [this](Run& R, int k) { process(0, R, k); }, [](Run& R) { R.finished.fetch_add(1); },
                       [this] { return !(poisoned[0].load() && stop(0)); });
  }
  void close() { queue.shutdown(); for (auto& t : threads) t.join(); }
};

static bool drained(const std::vector<std::shared_ptr<Run>>& runs, int ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    bool all = true;
    for (const auto& r : runs) all = all && r->finished.load() == 1;
    if (all) return true;
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > ms) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
}
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "retire harness: CHECK failed at line %d: %s\n", __LINE__, #c); exit(1); } } while (0)

int main() {
  // ---- 1. all but one THREADED worker poisoned, runs submitted and nobody waits: the healthy thread drains them ----
  {
    FakePool P(6);
    for (int w = 0; w <= 4; ++w) P.poisoned[(size_t)w] = true;   // the seat and threads 1 .. 4; thread 5 is healthy
    std::vector<std::shared_ptr<Run>> runs;
    for (int r = 0; r < 8; ++r) { runs.push_back(std::make_shared<Run>(40)); P.submit(runs.back()); }
    CHECK(drained(runs, 20000));
    for (const auto& r : runs) for (auto& h : r->hits) CHECK(h.load() == 1);
    // threads 1 .. 4 retire, each after its first item (one that never woke up in time has not yet); the seat was never occupied
    CHECK(P.retire.retired.load() >= 1 && P.retire.retired.load() <= 4 && P.retire_msgs.load() == P.retire.retired.load());
    for (const auto& r : runs) P.wait(r);                               // spent tickets return at once
    P.close();
  }
  // ---- 2. EVERY worker poisoned: the last claiming thread stays, the queue still drains, a wait comes back ----
  {
    FakePool P(5);
    for (auto& p : P.poisoned) p = true;
    std::vector<std::shared_ptr<Run>> runs;
    for (int r = 0; r < 6; ++r) { runs.push_back(std::make_shared<Run>(30)); P.submit(runs.back()); }
    CHECK(drained(runs, 20000));
    CHECK(P.retire.retired.load() >= 1 && P.retire.retired.load() <= 3);      // four threads: at most three retire, one goes on
    auto late = std::make_shared<Run>(25);
    P.submit(late);
    P.wait(late);                             // the seat's occupant steps back (a thread still claims); the run finishes all the same
    CHECK(late->finished.load() == 1);
    for (auto& h : late->hits) CHECK(h.load() == 1);
    CHECK(P.retire.retired.load() <= 4);      // + the seat, if its occupant got a turn at all
    { std::lock_guard<std::mutex> lk(P.retire.m); CHECK(P.retire.claiming_threads >= 1); }
    P.close();
  }
  // ---- 3. a pool of one: the seat is all there is, poisoned or not its occupant goes on ----
  {
    FakePool P(1);
    P.poisoned[0] = true;
    auto r = std::make_shared<Run>(20);
    P.submit(r);
    P.wait(r);
    CHECK(r->finished.load() == 1 && P.retire.retired.load() == 0);
    for (int k = 0; k < 20; ++k) CHECK(r->by[(size_t)k] == 0);
    P.close();
  }
  // ---- 4. nobody poisoned: nobody retires ----
  {
    FakePool P(4);
    std::vector<std::shared_ptr<Run>> runs;
    for (int r = 0; r < 5; ++r) { runs.push_back(std::make_shared<Run>(50)); P.submit(runs.back()); }
    for (const auto& r : runs) P.wait(r);
    CHECK(P.retire.retired.load() == 0);
    P.close();
  }
  printf("retire harness ok\n");
  return 0;
}
