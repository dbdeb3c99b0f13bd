// retire.h -- who stops taking chromosomes when a pool's contexts go bad.  Plain C++ (no HIP): tests/sanitize/retire_test.cpp
// drives it with fake flags.
//
// A context is POISONED when one of its waits gave up with kernels still queued (pipeline_internal.h: ctx_sync): every run on it
// fails at once.  Left in the rotation, such a worker would claim and fail chromosome after chromosome faster than the healthy
// workers can take them -- so it stops claiming.  But somebody must keep draining the queue, or runs queued with
// rsi_pool_submit would wait for ever:
//   * workers 1 .. W-1 have a thread each; worker 0 is a SEAT -- the context a caller uses while it waits (rsi_pool_wait), with
//     no thread of its own.  Only threads can be counted on: the seat is empty whenever nobody waits.
//   * a poisoned threaded worker retires unless it is the last threaded worker still claiming: that one goes on, failing what it
//     claims, so that the queue empties and every waiter gets its error;
//   * the seat's occupant stops helping when its context is poisoned and some thread still claims; with no thread left (or a
//     pool of one) it goes on, for the same reason.
// Every retirement is recorded (`retired`, and the pool's error text through `on_retire`).
#pragma once
#include <atomic>
#include <mutex>
#include <vector>

namespace rsip {

struct RetirePolicy {
  std::mutex m;
  std::vector<char> parked;   // per worker: stopped claiming
  int claiming_threads = 0;   // workers 1 .. W-1 that still claim
  std::atomic<int> retired{0};

  void reset(size_t nworkers) {
    std::lock_guard<std::mutex> lk(m);
    parked.assign(nworkers, 0);
    claiming_threads = nworkers > 0 ? (int)nworkers - 1 : 0;
    retired = 0;
  }
  // Worker w has just finished an item (or is about to help) and its context is `poisoned`: may it stop claiming?
  // `on_retire(w)` runs once per worker that does.
  template <class OnRetire>
  bool should_stop(size_t w, bool poisoned, OnRetire&& on_retire) {
    if (!poisoned) return false;
    std::lock_guard<std::mutex> lk(m);
    if (parked[w]) return true;
    if (w == 0) {                                   // the seat: nobody to hand over to unless a thread still claims
      if (claiming_threads == 0) return false;
    } else {
      if (claiming_threads <= 1) return false;      // the last claiming thread stays
      --claiming_threads;
    }
    parked[w] = 1;
    retired.fetch_add(1);
    on_retire(w);
    return true;
  }
};

}  // namespace rsip
