// ThreadSanitizer harness for RunQueue (rsicnv_amd/csrc/run_queue.h), the pool's queue of runs: eleven worker threads take items
// of queued runs, three client threads submit runs of different sizes and wait for them -- helping as a twelfth worker while
// they wait, as rsi_pool_wait does -- out of order and concurrently.  Checked: every item of every run is processed exactly
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

int main() {
  rsip::RunQueue<TestRun> q;
  std::atomic<long> processed(0), order_violations(0), young_help(0);
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
  for (int w = 0; w < 11; ++w) workers.emplace_back([&] {
    std::shared_ptr<TestRun> r; int k = 0;
    while (q.next(r, k)) { process(*r, k); q.item_done(r, finish); }
  });
  std::mutex helper;   // worker 0's context: one helping waiter at a time
  int failures = 0;
  std::mutex fail_mu;
  auto client = [&](int c) {
    for (int round = 0; round < 60; ++round) {
      std::vector<std::shared_ptr<TestRun>> mine;
      const int nruns = 1 + (round + c) % 3;
      for (int j = 0; j < nruns; ++j) {
        auto r = std::make_shared<TestRun>();
        r->nitems = (round * 7 + c * 3 + j) % 26;        // 0 .. 25 items: empty runs too
        r->result.assign((size_t)r->nitems, 0);
        r->touched.assign((size_t)r->nitems, 0);
        q.submit(r, finish, [](TestRun&) {});
        mine.push_back(r);
      }
      if ((round + c) % 2) std::swap(mine.front(), mine.back());   // wait out of order
      for (auto& r : mine) {
        auto found = q.find(r->id);
        if (found.get() != r.get()) { std::lock_guard<std::mutex> lk(fail_mu); ++failures; }
        if (helper.try_lock()) {
          std::shared_ptr<TestRun> h; int k = 0;
          while (q.try_next(r, h, k)) {
            if (h->id > r->id) young_help.fetch_add(1);
            process(*h, k);
            q.item_done(h, finish);
          }
          helper.unlock();
        }
        q.wait_done(r);
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
  for (int c = 0; c < 3; ++c) clients.emplace_back(client, c);
  for (auto& t : clients) t.join();
  q.shutdown();
  for (auto& t : workers) t.join();
  const bool ok = failures == 0 && young_help.load() == 0 && q.active.empty() && q.unwaited.empty();
  printf("runs checked by 3 clients, %ld items processed, %d failures, helper on younger runs %ld, queue empty at the end %d\n",
         processed.load(), failures, young_help.load(), (int)(q.active.empty() && q.unwaited.empty()));
  if (!ok) return 1;
  printf("queue harness ok\n");
  return 0;
}
