// run_queue.h -- the pool's queue of runs.  Plain C++ (no HIP): tests/sanitize/queue_tsan.cpp drives it from a dozen
// threads under ThreadSanitizer.
//
// A run is one call's worth of items (chromosomes).  Runs queue up in submission order; whoever asks for work gets the next
// unclaimed item of the OLDEST run that has one, so a worker that finds nothing left in one run starts on the next: consecutive
// samples overlap, and the last items of a run do not leave the other workers idle.  A run is finished when all of its items
// have been reported done; `finish(run)` is called once, under the queue's mutex, by the thread that reports the last one,
// before any waiter of that run wakes up.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <vector>

namespace rsip {

struct RunBase {
  uint64_t id = 0;
  int nitems = 0;
  int claimed = 0, completed = 0;   // under the queue's mutex
  bool done = false;
};

template <class Run>   // Run derives from RunBase
struct RunQueue {
  std::mutex m;
  std::condition_variable work_cv, done_cv;
  std::deque<std::shared_ptr<Run>> active;      // runs with unfinished items, oldest first
  std::vector<std::shared_ptr<Run>> unwaited;   // submitted, wait() not yet returned
  uint64_t next_id = 1;
  bool quit = false;
  bool helper_busy = false;                     // a waiting caller is processing an item with the helpers' shared resource

  // under m: the next item to work on, from the oldest run that has one (no younger than `upto`; 0 = any)
  bool claim_locked(std::shared_ptr<Run>& r, int& k, uint64_t upto) {
    for (auto& a : active) {
      if (upto && a->id > upto) break;
      if (a->claimed < a->nitems) { r = a; k = a->claimed++; return true; }
    }
    return false;
  }
  // Queues the run and returns its ticket.  `before_publish(run)` runs under the mutex before any worker can see the run.
  // caller_first: the submitting thread is about to wait_helping() for this run and the run is a single item -- the workers are
  // not woken, the caller takes the item itself on the helpers' context (the same one every time: its buffers are allocated and
  // warm, where whichever of a dozen woken workers wins the race may never have seen this kind of chromosome).  A caller that
  // cannot take it (the helper's seat is occupied, or helping is off) wakes the workers from wait_helping().
  template <class Finish, class Before>
  uint64_t submit(const std::shared_ptr<Run>& r, Finish&& finish, Before&& before_publish, bool caller_first = false) {
    {
      std::lock_guard<std::mutex> lk(m);
      r->id = next_id++;
      before_publish(*r);
      unwaited.push_back(r);
      if (r->nitems == 0) { finish(*r); r->done = true; }
      else active.push_back(r);
    }
    if (!(caller_first && r->nitems == 1)) work_cv.notify_all();
    return r->id;
  }
  // under m: an item of the runs up to `upto` is still unclaimed
  bool unclaimed_locked(uint64_t upto) const {
    for (const auto& a : active) {
      if (upto && a->id > upto) break;
      if (a->claimed < a->nitems) return true;
    }
    return false;
  }
  bool empty_locked() const { return active.empty(); }
  // Blocks until there is an item to work on (true) or the queue is shut down (false).
  bool next(std::shared_ptr<Run>& r, int& k) {
    std::unique_lock<std::mutex> lk(m);
    work_cv.wait(lk, [&] { return quit || claim_locked(r, k, 0); });
    return !quit;
  }
  // Without blocking: an item of the runs up to `upto`, unless that run is finished already.
  bool try_next(const std::shared_ptr<Run>& mine, std::shared_ptr<Run>& r, int& k) {
    std::lock_guard<std::mutex> lk(m);
    return !mine->done && claim_locked(r, k, mine->id);
  }
  // under m: one item of the run has been processed; the last one closes the run.
  template <class Finish>
  void complete_locked(const std::shared_ptr<Run>& r, Finish&& finish) {
    if (++r->completed < r->nitems) return;
    finish(*r);
    r->done = true;
    for (auto it = active.begin(); it != active.end(); ++it) if (it->get() == r.get()) { active.erase(it); break; }
    done_cv.notify_all();
  }
  template <class Finish>
  void item_done(const std::shared_ptr<Run>& r, Finish&& finish) {
    std::lock_guard<std::mutex> lk(m);
    complete_locked(r, finish);
  }
  std::shared_ptr<Run> find(uint64_t ticket) {
    std::lock_guard<std::mutex> lk(m);
    for (auto& r : unwaited) if (r->id == ticket) return r;
    return nullptr;
  }
  // Blocks until the run is finished; the ticket is spent afterwards.
  void wait_done(const std::shared_ptr<Run>& mine) {
    std::unique_lock<std::mutex> lk(m);
    done_cv.wait(lk, [&] { return mine->done; });
    for (auto it = unwaited.begin(); it != unwaited.end(); ++it) if (it->get() == mine.get()) { unwaited.erase(it); break; }
  }
  // Blocks until the run is finished, and works while it waits: the caller takes items of its own run and of the runs ahead
  // of it (never of a younger run) and hands them to `process(run, k)`.  The helpers share ONE resource (the pool's worker-0
  // context), so one helper processes at a time; the others sleep and are woken when the helper's item is done -- whoever
  // wakes first is the next helper.  A waiter therefore never sleeps while an item it may take is unclaimed and no helper
  // is at work: with no background workers at all, any number of waiting threads still drain the queue (round 3's form
  // tried the helper role once and then slept for good, which left the second of two waiters asleep next to its unclaimed
  // run).  `may_help()` = false (evaluated under the mutex) turns the caller into a plain waiter.
  template <class Process, class Finish, class MayHelp>
  void wait_helping(const std::shared_ptr<Run>& mine, Process&& process, Finish&& finish, MayHelp&& may_help) {
    std::unique_lock<std::mutex> lk(m);
    while (!mine->done) {
      std::shared_ptr<Run> r;
      int k = 0;
      if (!helper_busy && may_help() && claim_locked(r, k, mine->id)) {
        if (r.get() != mine.get() && mine->claimed < mine->nitems) work_cv.notify_all();   // an older run's item first: mine is the workers' meanwhile
        helper_busy = true;
        lk.unlock();
        process(*r, k);
        lk.lock();
        helper_busy = false;
        complete_locked(r, finish);
        done_cv.notify_all();     // the helper's seat is free: another waiter may have claimable items
        continue;
      }
      if (unclaimed_locked(mine->id)) work_cv.notify_all();   // not mine to take right now: the workers' (a caller-first run never woke them)
      done_cv.wait(lk);
    }
    for (auto it = unwaited.begin(); it != unwaited.end(); ++it) if (it->get() == mine.get()) { unwaited.erase(it); break; }
  }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(m); quit = true; }
    work_cv.notify_all();
  }
};

}  // namespace rsip
