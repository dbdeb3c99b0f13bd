// synth_host.cpp -- table construction + host generator for the synthetic chromosomes
// (include/rsi_synth.h).  Compiled with -ffp-contract=off so that the tables are the same
// whatever CPU runs this.
#include <string.h>
#include <thread>
#include <string>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "synth_tables.h"

namespace {

const double kCnFactor[RSI_CN_CLASSES] = {1.0, 0.0, 0.5, 1.5, 2.0};

// Unnormalised weights w[0..K] of a count distribution given the ratio w[k+1]/w[k] = num(k)/den(k),
// cut where the upper tail is below 1e-25 of the peak; then thresholds floor(cdf * 2^53).
template <class Ratio>
void build_class(Ratio ratio, bool degenerate, std::vector<uint64_t>& out) {
  out.clear();
  if (degenerate) return;   // always 0
  std::vector<double> w;
  w.push_back(1.0);
  double peak = 1.0;
  for (int k = 0; k < 200000; ++k) {
    double nxt = w[k] * ratio(k);
    if (nxt > peak) peak = nxt;
    if (nxt < w[k] && nxt < 1e-25 * peak) break;
    w.push_back(nxt);
  }
  double total = 0.0;
  for (size_t k = 0; k < w.size(); ++k) total += w[k];
  const double two53 = 9007199254740992.0;
  double acc = 0.0;
  // thresholds for values 0..K-1; value K is "everything above"
  for (size_t k = 0; k + 1 < w.size(); ++k) {
    acc += w[k];
    double c = acc / total;
    if (c > 1.0) c = 1.0;
    out.push_back((uint64_t)(c * two53));
  }
}

}  // namespace

void synth_build_tables(const rsi_synth_spec& spec, SynthTables& T) {
  T.wave.resize(RSI_SYNTH_WAVE);
  for (int k = 0; k < RSI_SYNTH_WAVE; ++k) {
    double t = (double)k / (double)RSI_SYNTH_WAVE;
    double s;
    if (t < 0.5) s = 16.0 * t * (0.5 - t);
    else { double u = t - 0.5; s = -16.0 * u * (0.5 - u); }
    double p = 0.41 + 0.10 * s;
    T.wave[k] = (uint32_t)(p * 4294967296.0);
  }
  T.gc_levels = spec.model == 0 ? 1 : RSI_SYNTH_GC_LEVELS;
  T.thr.clear();
  T.off.clear();
  std::vector<uint64_t> cls;
  for (int g = 0; g < T.gc_levels; ++g) {
    double gcfac = spec.model == 0 ? 1.0 : (0.7 + 0.6 * ((double)g / 201.0));
    for (int c = 0; c < RSI_CN_CLASSES; ++c) {
      double mu = spec.mean * gcfac * kCnFactor[c];
      T.off.push_back((int32_t)T.thr.size());
      if (spec.model == 0) {
        build_class([mu](int k) { return mu / (double)(k + 1); }, mu <= 0.0, cls);
      } else {
        double size = spec.nb_size;
        double q = mu / (mu + size);
        build_class([q, size](int k) { return ((double)k + size) / (double)(k + 1) * q; }, mu <= 0.0, cls);
      }
      T.thr.insert(T.thr.end(), cls.begin(), cls.end());
    }
  }
  T.off.push_back((int32_t)T.thr.size());
}

extern "C" int rsi_synth_generate_host(const rsi_synth_spec* spec, uint8_t* fasta, int32_t* depth) {
  if (!spec || spec->n <= 0 || !fasta || !depth) return -1;
  if (spec->model != 0 && spec->model != 1) return -2;
  SynthTables T;
  synth_build_tables(*spec, T);
  const int64_t n = spec->n;
  const rsi_synth_interval* ev = reinterpret_cast<const rsi_synth_interval*>(spec->events);
  const rsi_synth_interval* nr = reinterpret_cast<const rsi_synth_interval*>(spec->nruns);
  const rsi_synth_interval* lo = reinterpret_cast<const rsi_synth_interval*>(spec->lower);
  const uint64_t seed_fa = rsi_mix64(spec->seed ^ 0xFA57A000ULL);
  const uint64_t seed_rd = rsi_mix64(spec->seed ^ 0xDE97B000ULL);

  unsigned hw = std::thread::hardware_concurrency();
  int nt = hw == 0 ? 1 : (int)(hw > 8 ? 8 : hw);
  if (n < 1000000) nt = 1;
  auto run = [&](auto fn) {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) {
      int64_t a = n * t / nt, b = n * (t + 1) / nt;
      th.emplace_back([=]() { fn(a, b); });
    }
    for (auto& x : th) x.join();
  };
  run([&](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i)
      fasta[i] = rsi_synth_base(seed_fa, i, T.wave.data(), nr, spec->n_nruns, lo, spec->n_lower);
  });
  run([&](int64_t a, int64_t b) {
    int g = 0;
    if (spec->model == 1) {   // window count for position a
      int64_t w0 = a - 100 < 0 ? 0 : a - 100, w1 = a + 100 >= n ? n - 1 : a + 100;
      for (int64_t j = w0; j <= w1; ++j) g += rsi_is_gc(fasta[j]);
    }
    for (int64_t i = a; i < b; ++i) {
      int cn = RSI_CN_1X;
      int e = rsi_find_interval(ev, spec->n_events, i);
      if (e >= 0) cn = ev[e].code;
      int32_t d;
      if (fasta[i] == 'N') d = 0;
      else d = rsi_synth_depth(seed_rd, i, (spec->model == 1 ? g : 0) * RSI_CN_CLASSES + cn, T.thr.data(), T.off.data());
      depth[i] = d;
      if (spec->model == 1) {   // slide to i+1: window [i-99, i+101]
        if (i - 100 >= 0) g -= rsi_is_gc(fasta[i - 100]);
        if (i + 101 < n) g += rsi_is_gc(fasta[i + 101]);
      }
    }
  });
  return 0;
}

// Test / bench plumbing: the "pos<TAB>depth" text of samtools mpileup | cut -f2,4 for depth[n], and the FASTA file (60
// bases per line) with its .fai index for one chromosome -- written fast enough (hand-rolled digits, big buffers) that a
// 60 Mb chromosome's 700 MB of text take a second or two instead of a minute of Python formatting.
extern "C" int rsi_synth_write_depth_text(const char* path, const int32_t* depth, int64_t n) {
  if (!path || !depth || n <= 0) return -1;
  FILE* f = fopen(path, "wb");
  if (!f) return -2;
  const size_t kBuf = size_t(16) << 20;
  std::vector<char> buf(kBuf + 64);
  size_t used = 0;
  auto put_int = [&](long long v) {
    char tmp[24];
    int k = 0;
    unsigned long long u = v < 0 ? (unsigned long long)(-v) : (unsigned long long)v;
    do { tmp[k++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) buf[used++] = '-';
    while (k) buf[used++] = tmp[--k];
  };
  const char head[] = "#pos depth\n";
  memcpy(buf.data(), head, sizeof(head) - 1); used = sizeof(head) - 1;
  for (int64_t i = 0; i < n; ++i) {
    put_int(i + 1); buf[used++] = '\t'; put_int(depth[i]); buf[used++] = '\n';
    if (used >= kBuf) { if (fwrite(buf.data(), 1, used, f) != used) { fclose(f); return -3; } used = 0; }
  }
  if (used && fwrite(buf.data(), 1, used, f) != used) { fclose(f); return -3; }
  return fclose(f) == 0 ? 0 : -3;
}

extern "C" int rsi_synth_write_fasta(const char* path, const char* chrom, const uint8_t* fasta, int64_t n) {
  if (!path || !chrom || !fasta || n <= 0) return -1;
  FILE* f = fopen(path, "wb");
  if (!f) return -2;
  const std::string head = std::string(">") + chrom + "\n";
  fwrite(head.data(), 1, head.size(), f);
  std::vector<char> line(61 * 16384);
  for (int64_t i = 0; i < n;) {
    size_t used = 0;
    for (int l = 0; l < 16384 && i < n; ++l) {
      const int64_t k = n - i < 60 ? n - i : 60;
      memcpy(line.data() + used, fasta + i, (size_t)k); used += (size_t)k; line[used++] = '\n';
      i += k;
    }
    if (fwrite(line.data(), 1, used, f) != used) { fclose(f); return -3; }
  }
  if (fclose(f) != 0) return -3;
  FILE* fai = fopen((std::string(path) + ".fai").c_str(), "w");
  if (!fai) return -2;
  fprintf(fai, "%s\t%lld\t%zu\t60\t61\n", chrom, (long long)n, head.size());
  return fclose(fai) == 0 ? 0 : -3;
}
