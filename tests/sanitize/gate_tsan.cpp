// ThreadSanitizer harness for GpuGate (rsicnv_amd/csrc/gate.h), the one piece of the pool's host threading that is plain
// C++: a dozen threads take per-base turns (at most max_streamers at once) and shared sections in both schedules -- the
// default, where shared sections run freely, and the isolated one, where a per-base turn excludes them -- while counters
// check the invariants.  CPU only.   g++ -fsanitize=thread ... && tests/sanitize/gate_tsan
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include "../../rsicnv_amd/csrc/gate.h"

int main() {
  int failures = 0;
  for (int isolate = 0; isolate < 2; ++isolate) {
    for (int maxs = 1; maxs <= 3; ++maxs) {
      rsip::GpuGate gate;
      gate.max_streamers = maxs;
      std::atomic<int> streaming(0), sharing(0), worst_streaming(0), overlap(0);
      long plain = 0;   // written only while holding an exclusive turn at max_streamers == 1: a data race here is a gate bug
      auto worker = [&](int id) {
        for (int it = 0; it < 400; ++it) {
          gate.lock(isolate != 0);
          const int s = streaming.fetch_add(1) + 1;
          int w = worst_streaming.load();
          while (s > w && !worst_streaming.compare_exchange_weak(w, s)) {}
          if (isolate && sharing.load() != 0) overlap.fetch_add(1);
          if (maxs == 1 || isolate) ++plain;
          if ((it + id) % 7 == 0) std::this_thread::sleep_for(std::chrono::microseconds(20));
          streaming.fetch_sub(1);
          gate.unlock();
          if (isolate) {   // bin-level sections of the isolated schedule
            gate.lock_shared();
            sharing.fetch_add(1);
            if (streaming.load() != 0) overlap.fetch_add(1);
            if ((it + id) % 5 == 0) std::this_thread::yield();
            sharing.fetch_sub(1);
            gate.unlock_shared();
          }
        }
      };
      std::vector<std::thread> th;
      for (int t = 0; t < 12; ++t) th.emplace_back(worker, t);
      for (auto& t : th) t.join();
      const int limit = isolate ? 1 : maxs;
      const bool ok = worst_streaming.load() <= limit && overlap.load() == 0 && ((maxs != 1 && !isolate) || plain == 12L * 400);
      printf("isolate %d max_streamers %d: most per-base turns at once %d (limit %d), overlaps %d%s\n", isolate, maxs, worst_streaming.load(),
             limit, overlap.load(), ok ? "" : "  <-- FAILED");
      if (!ok) ++failures;
    }
  }
  if (failures) return 1;
  printf("gate harness ok\n");
  return 0;
}
