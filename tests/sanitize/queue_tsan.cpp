// ThreadSanitizer harness for RunQueue (rsicnv_amd/csrc/run_queue.h), the pool's queue of runs: eleven worker threads take items
// of queued runs, three client threads submit runs of different sizes and wait for them -- helping as a twelfth worker while
// they wait, as rsi_pool_wait does (RunQueue::wait_helping) -- out of order and concurrently; then the same with NO worker
// threads at all (a pool of one context: the waiting callers are the only workers, one at a time).  Checked: every item of every run is processed exactly
// once, `finish` runs once per run before its waiter returns and sees all of the run's results, items are claimed oldest run
// first, a helper never touches a run younger than its own, and ThreadSanitizer sees no race.  CPU only.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include "../../rsicnv_amd/csrc/run_queue.h"

struct TestRun : rsip::RunBase {
  std::vector<int> result;        // written by whoever processes an item: plain memory, ordered by the queue's mutex alone
  std::vector<int> touched;       // how often an item was processed
  long sum_at_finish = -1;        // what finish() saw
  int finished = 0;
};

static int drive(int nworkers, int nclients, int rounds) {
  rsip::RunQueue<TestRun> q;
  std::atomic<long> processed(0), young_help(0);
  std::atomic<int> helpers_inside(0), helper_overlap(0);
  std::atomic<uint64_t> oldest_unfinished(1);
  auto process = [&](TestRun& r, int k) {
    r.result[(size_t)k] = (int)r.id * 1000 + k;
    ++r.touched[(size_t)k];
    if ((k + (int)r.id) % 5 == 0) std::this_thread::sleep_for(std::chrono::microseconds(30));
    processed.fetch_add(1);
  };
  auto finish = [&](TestRun& r) {
    long s = 0;
    for (int v : r.result) s += v;
    r.sum_at_finish = s;
    ++r.finished;
  };
  std::vector<std::thread> workers;
  for (int w = 0; w < nworkers; ++w) workers.emplace_back([&] {
    std::shared_ptr<TestRun> r; int k = 0;
    while (q.next(r, k)) { process(*r, k); q.item_done(r, finish); }
  });
  int failures = 0;
  std::mutex fail_mu;
  auto client = [&](int c) {
    for (int round = 0; round < rounds; ++round) {
      std::vector<std::shared_ptr<TestRun>> mine;
      const int nruns = 1 + (round + c) % 3;
      for (int j = 0; j < nruns; ++j) {
        auto r = std::make_shared<TestRun>();
        r->nitems = (round * 7 + c * 3 + j) % 26;        // 0 .. 25 items: empty runs too
        r->result.assign((size_t)r->nitems, 0);
        r->touched.assign((size_t)r->nitems, 0);
        q.submit(r, finish, [](TestRun&) {}, (round + j) % 2 == 0);   // every other run "caller first": a single item does not wake the workers
        mine.push_back(r);
      }
      if ((round + c) % 2) std::swap(mine.front(), mine.back());   // wait out of order
      for (auto& r : mine) {
        auto found = q.find(r->id);
        if (found.get() != r.get()) { std::lock_guard<std::mutex> lk(fail_mu); ++failures; }
        const uint64_t my_id = r->id;
        q.wait_helping(r, [&](TestRun& h, int k) {
          if (helpers_inside.fetch_add(1) != 0) helper_overlap.fetch_add(1);   // worker 0's context: one helper at a time
          if (h.id > my_id) young_help.fetch_add(1);
          process(h, k);
          helpers_inside.fetch_sub(1);
        }, finish, [&] { return nworkers == 0 || (c + round) % 3 != 0; });   // with workers around, a caller that may not help now and then
        long want = 0;
        bool once = true;
        for (int k = 0; k < r->nitems; ++k) { want += (long)r->id * 1000 + k; once = once && r->touched[(size_t)k] == 1; }
        if (!(r->done && r->finished == 1 && r->sum_at_finish == want && once && r->completed == r->nitems)) {
          std::lock_guard<std::mutex> lk(fail_mu);
          ++failures;
        }
        if (q.find(r->id)) { std::lock_guard<std::mutex> lk(fail_mu); ++failures; }   // the ticket is spent
      }
    }
  };
  std::vector<std::thread> clients;
  for (int c = 0; c < nclients; ++c) clients.emplace_back(client, c);
  for (auto& t : clients) t.join();
  q.shutdown();
  for (auto& t : workers) t.join();
  const bool ok = failures == 0 && young_help.load() == 0 && helper_overlap.load() == 0 && q.active.empty() && q.unwaited.empty();
  printf("%d workers, %d clients: %ld items processed, %d failures, helper on younger runs %ld, two helpers at once %d, queue empty at the end %d\n",
         nworkers, nclients, processed.load(), failures, young_help.load(), helper_overlap.load(), (int)(q.active.empty() && q.unwaited.empty()));
  return ok ? 0 : 1;
}

int main() {
  if (drive(11, 3, 60)) return 1;
  // no background workers (rsi_pool_create(nworkers = 1)): two and four waiting callers are all the workers there are --
  // round 3's wait slept for good next to its own unclaimed run here
  if (drive(0, 2, 60)) return 1;
  if (drive(0, 4, 40)) return 1;
  if (drive(1, 3, 40)) return 1;
  printf("queue harness ok\n");
  return 0;
}
