// gate.h -- GpuGate, the turnstile of a pool's per-base phases.  Plain C++ (no HIP): tests/sanitize/gate_tsan.cpp drives it
// from a dozen threads under ThreadSanitizer.
#pragma once
#include <atomic>
#include <condition_variable>
#include <mutex>

namespace rsip {

// One GPU, several workers.  The per-base phase of a chromosome is HBM-bound, so running many of
// them at once gains nothing and costs L2 locality: at most `max_streamers` are in flight (four: the
// phase has a host round trip -- N-run list, cap median, bin statistics -- and its big kernels end in one-workgroup tails;
// the others' kernels fill both.  Two was the measured optimum while a pool ran one genome at a time; with genomes queued
// back to back, 16 workers and three phases took 14.2-14.9 ms per genome against 15.2 for 12 and two; with the per-base
// kernels of round 5, 20 workers and four phases take 12.2 ms against 12.5 for 16 and three, tools/gpu_sched_sweep2.sh).
// Bin-level work of other chromosomes overlaps freely.  With RSI_HOT_ISOLATE_STREAMING=1
// a per-base phase runs alone on the chip (bin-level sections wait, waiting streamers hold back new
// sharers): every streaming launch is then a clean roofline sample, at about 20 % less throughput.
struct GpuGate {
  std::mutex m;
  std::condition_variable cv;
  int sharers = 0, streamers_waiting = 0, streaming = 0;
  std::atomic<bool> few_chromosomes{false};   // set per submitted run: so few chromosomes that latency, not sharing, decides (pipeline.hip, candidate tests)
  std::atomic<int> in_flight{0};              // chromosomes the pool's workers are processing right now
  // Latency, not sharing, decides: a run of a few chromosomes, or the last few of a genome with nothing queued behind it.
  bool lonely() const { return few_chromosomes.load() || in_flight.load() <= 4; }
  int max_streamers = 4;   // per-base phases in flight: the others fill the host gaps (syncs, small decisions) and one-workgroup tails of one
  void lock_shared() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return streaming == 0 && streamers_waiting == 0; }); ++sharers; }
  void unlock_shared() { { std::lock_guard<std::mutex> lk(m); --sharers; } cv.notify_all(); }
  void lock(bool exclude_sharers) {
    std::unique_lock<std::mutex> lk(m);
    ++streamers_waiting;
    cv.wait(lk, [&] { return exclude_sharers ? (streaming == 0 && sharers == 0) : streaming < max_streamers; });
    --streamers_waiting;
    ++streaming;
  }
  void unlock() { { std::lock_guard<std::mutex> lk(m); --streaming; } cv.notify_all(); }
};

}  // namespace rsip
