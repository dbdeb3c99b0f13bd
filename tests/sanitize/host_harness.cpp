// host_harness.cpp -- CPU-only driver of the product's HOST code for the sanitizer build (tests/sanitize/Makefile: ASan +
// UBSan; test infrastructure, never shipped).  It runs, on seeded synthetic chromosomes,
//   * hostmath.h's histogram quantiles against the oracle's (the CPU restatement, pinned to the reference elsewhere),
//   * host_calls.cpp's candidate stages (block tests, sharpening, neighbourhood tests, merge, final filters) on their
//     host path -- the depth in host memory, no device tester -- from the oracle's bin arrays and segments, and compares
//     blocks / raw calls / final calls with the oracle's,
//   * bam_host.cpp's BGZF / BAM / BAI reader and the read-pair annotation on a BAM given on the command line, then on
//     truncated and bit-flipped copies of it (errors are fine, memory errors are not).
// Exit status 0 = everything agreed and the sanitizers stayed quiet.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "../../include/rsi_hot.h"
#include "../../include/rsi_synth.h"
#include "../../rsicnv_amd/csrc/bam_host.h"
#include "../../rsicnv_amd/csrc/host_calls.h"
#include "../../rsicnv_amd/csrc/hostmath.h"

extern "C" {   // oracle/rsi_oracle.cpp
struct orc_params { int32_t m, gcadjust, trans, merge, maxchkbp, debug; double cap, epsilon, threshold, chklen, minmlen, buffer, p; };
struct orc_call { int32_t start, end, type, geno, status, length, qscore, pad; double score, p1, cnvmed, cnvsd, cnviqr, refmed, refsd, refiqr; };
void* orc_create(void);
void orc_destroy(void* h);
void orc_default_params(orc_params* p);
int orc_run(void* h, const orc_params* p, const int32_t* depth, const uint8_t* fasta, int32_t n, int32_t keep_snapshots);
int64_t orc_get_i32(void* h, const char* name, int32_t* out, int64_t cap);
int64_t orc_get_f64(void* h, const char* name, double* out, int64_t cap);
int orc_get_calls(void* h, const char* which, orc_call* out, int32_t cap);
double orc_median_i32(const int32_t* x, int64_t n);
double orc_median_f32(const float* x, int64_t n);
double orc_median_f64(const double* x, int64_t n);
}

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); ++g_fail; } } while (0)

static std::vector<int32_t> geti(void* h, const char* name) {
  const int64_t k = orc_get_i32(h, name, nullptr, 0);
  std::vector<int32_t> v((size_t)(k > 0 ? k : 0));
  if (k > 0) orc_get_i32(h, name, v.data(), k);
  return v;
}
static std::vector<orc_call> getc(void* h, const char* which) {
  const int k = orc_get_calls(h, which, nullptr, 0);
  std::vector<orc_call> v((size_t)(k > 0 ? k : 0));
  if (k > 0) orc_get_calls(h, which, v.data(), k);
  return v;
}

static void quantile_checks() {
  std::mt19937_64 rng(7);
  for (int n : {1, 2, 3, 4, 5, 31, 100, 1001, 50000}) {
    std::vector<int> xi((size_t)n); std::vector<float> xf((size_t)n); std::vector<double> xd((size_t)n);
    std::poisson_distribution<int> po(30); std::gamma_distribution<double> ga(9.0, 3.3);
    for (int i = 0; i < n; ++i) { xi[(size_t)i] = po(rng); xf[(size_t)i] = (float)ga(rng); xd[(size_t)i] = (double)xf[(size_t)i] * 1.37; }
    CHECK(rsih::grid_quantiles(xi.data(), (size_t)n).med == orc_median_i32(xi.data(), n), "int median, n=%d", n);
    CHECK(rsih::grid_quantiles(xf.data(), (size_t)n).med == orc_median_f32(xf.data(), n), "float median, n=%d", n);
    CHECK(rsih::grid_quantiles(xd.data(), (size_t)n).med == orc_median_f64(xd.data(), n), "double median, n=%d", n);
    // device-style integer histogram -> quantiles
    int lo = xi[0], hi = xi[0];
    for (int v : xi) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    std::vector<uint64_t> h((size_t)hi + 1, 0);
    for (int v : xi) ++h[(size_t)v];
    rsih::Quantiles q;
    CHECK(rsih::hist_quantiles_int(h.data(), h.size(), (uint64_t)n, q) && q.med == orc_median_i32(xi.data(), n), "histogram median, n=%d", n);
  }
}

static bool same_calls(const std::vector<rsih::Candidate>& a, const std::vector<orc_call>& b, const char* what) {
  if (a.size() != b.size()) { CHECK(false, "%s: %zu calls, oracle %zu", what, a.size(), b.size()); return false; }
  for (size_t i = 0; i < a.size(); ++i) {
    const bool ok = a[i].start == b[i].start && a[i].end == b[i].end && a[i].type == b[i].type && a[i].status == b[i].status &&
                    a[i].length == b[i].length && fabs(a[i].p1 - b[i].p1) <= 1e-9 * fabs(b[i].p1) + 1e-300 &&
                    fabs(a[i].cnvmed - b[i].cnvmed) <= 1e-9 * fabs(b[i].cnvmed) && fabs(a[i].refmed - b[i].refmed) <= 1e-9 * fabs(b[i].refmed);
    CHECK(ok, "%s: call %zu differs (%d-%d type %d vs %d-%d type %d)", what, i, a[i].start, a[i].end, a[i].type, b[i].start, b[i].end, b[i].type);
    if (!ok) return false;
  }
  return true;
}

static void candidate_stage_case(uint64_t seed, int n, int model, const orc_params& P, int deep_bimodal = 0) {
  // a chromosome with N runs at the ends, one gap and a few events (layout as tests/conftest.py's plans)
  std::vector<rsi_synth_interval_c> nruns = {{0, 4000, 0, 0}, {n / 2, n / 2 + 6000, 0, 0}, {n - 4000, n, 0, 0}};
  std::vector<rsi_synth_interval_c> events;
  const int codes[5] = {2, 3, 1, 4, 2};
  for (int e = 0; e < 5; ++e) { const int64_t b = 20000 + (int64_t)e * (n / 6); events.push_back({b, b + 3000 + 4000 * e, codes[e], 0}); }
  for (auto& ev : events) if (ev.beg < n / 2 + 6000 && ev.end > n / 2) { ev.beg += 20000; ev.end += 20000; }
  rsi_synth_spec spec;
  memset(&spec, 0, sizeof(spec));
  spec.seed = seed; spec.n = n; spec.model = model; spec.mean = 30.0; spec.nb_size = 10.0;
  spec.events = events.data(); spec.n_events = (int)events.size(); spec.nruns = nruns.data(); spec.n_nruns = (int)nruns.size();
  std::vector<uint8_t> fasta((size_t)n); std::vector<int32_t> depth((size_t)n);
  CHECK(rsi_synth_generate_host(&spec, fasta.data(), depth.data()) == 0, "generator");
  if (deep_bimodal) {   // 300x with the second half three times as deep: ~1800 segments, hundreds of candidates, neighbour chains
                        // cut at 48, a candidate whose first edge refinement leaves end < start (the reference indexes its
                        // vector out of range there; oracle and library search the entries that exist -- this case under
                        // AddressSanitizer is what keeps both honest), thinned neighbourhoods of a 170 kb candidate
    for (int i = 0; i < n; ++i) if (depth[(size_t)i] > 0) depth[(size_t)i] = depth[(size_t)i] * 10 + (int)((seed + (uint64_t)i * 2654435761u) % 10);
    for (int i = n * 9 / 20; i < n; ++i) if (depth[(size_t)i] > 0) depth[(size_t)i] = depth[(size_t)i] * 3 + 1;
  }
  void* O = orc_create();
  orc_run(O, &P, depth.data(), fasta.data(), n, 0);
  const std::vector<int32_t> rdc = geti(O, "rd_concat"), medint = geti(O, "binmedint"), noncode = geti(O, "noncode");
  const bool med = P.trans == 1;
  const std::vector<int32_t> st2 = geti(O, med ? "med_status2" : "nb_status2");
  const std::vector<orc_call> segs_o = getc(O, med ? "segs_med" : "segs_nb");
  double chrom[4];
  orc_get_f64(O, "chrom", chrom, 4);
  rsih::CallerInput in;
  memset(&in.P, 0, sizeof(in.P));
  in.P.m = P.m; in.P.gcadjust = P.gcadjust; in.P.trans = P.trans; in.P.merge = P.merge; in.P.maxchkbp = P.maxchkbp; in.P.debug = 0;
  in.P.cap = P.cap; in.P.epsilon = P.epsilon; in.P.threshold = P.threshold; in.P.chklen = P.chklen; in.P.minmlen = P.minmlen; in.P.buffer = P.buffer; in.P.p = P.p;
  in.RDmedian = chrom[0]; in.RDsd = chrom[1]; in.ncompact = (int64_t)rdc.size();
  std::vector<rsih::Region> regions;
  for (size_t i = 0; i + 1 < noncode.size(); i += 2) regions.push_back({noncode[i], noncode[i + 1]});
  in.noncode = &regions;
  std::vector<int> mi(medint.begin(), medint.end()), status(st2.begin(), st2.end());
  in.binmedint = rsih::IntSpan(mi);
  std::vector<rsih::Candidate> segs;
  for (const orc_call& c : segs_o) { rsih::Candidate k; k.start = c.start; k.end = c.end; k.type = c.type; k.score = c.score; segs.push_back(k); }
  rsih::test_block_segments(in, status, segs);                        // areblockscnv, rsi.cpp:1847
  rsih::DepthPager pager(rdc.data(), (int64_t)rdc.size());
  std::vector<rsih::Candidate> blocks, raw, kept;
  rsih::call_from_segments(in, segs, pager, blocks, raw, kept);       // rsi.cpp:1860-1931 + sd_filters
  same_calls(blocks, getc(O, "blocks"), "blocks");
  same_calls(raw, getc(O, "calls_raw"), "calls_raw");
  same_calls(kept, getc(O, "calls"), "calls");
  CHECK(!raw.empty(), "the case should call something (seed %llu)", (unsigned long long)seed);
  if (deep_bimodal) CHECK(raw.size() > 100, "the bimodal case should be crowded (%zu raw calls)", raw.size());
  orc_destroy(O);
}

static void bam_checks(const char* path) {
  std::ifstream f(path, std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  CHECK(!raw.empty(), "cannot read %s", path);
  auto walk = [&](const std::string& p, bool must_work) {
    std::string err;
    rsih::BamFile bam;
    std::vector<std::pair<std::string, int64_t>> refs;
    uint64_t first = 0;
    if (!bam.open(p, err) || !bam.read_header(refs, first, err)) { CHECK(!must_work, "header of %s: %s", p.c_str(), err.c_str()); return; }
    rsih::BamReader rd(bam);
    if (!rd.seek(first, err)) { CHECK(!must_work, "seek: %s", err.c_str()); return; }
    rsih::BamRecord r;
    long n = 0;
    int rc;
    while ((rc = rd.next(r, err)) == 1 && n < 2000000) ++n;
    CHECK(!must_work || (rc == 0 && n > 0), "record walk of %s ended with %d after %ld records: %s", p.c_str(), rc, n, err.c_str());
    for (size_t t = 0; t < refs.size() && t < 3; ++t) {
      rsih::PairSample ps;
      (void)rsih::bam_pair_sample(bam, p + ".bai", (int)t, refs[t].second, 1000, refs[t].second, ps, err);
      std::vector<rsih::CallSpan> calls = {{2000, 9000, 0, -1, -1.0}, {(int)(refs[t].second / 2), (int)(refs[t].second / 2) + 5000, 1, -1, -1.0}};
      (void)rsih::bam_annotate_calls(bam, p + ".bai", (int)t, ps, calls, err);
      uint64_t v = 0;
      (void)rsih::bai_first_offset(p + ".bai", (int)t, v);
    }
  };
  walk(path, true);
  std::mt19937_64 rng(99);
  const std::string tmp = std::string(path) + ".mangled";
  for (int trial = 0; trial < 24; ++trial) {
    std::vector<char> bad = raw;
    if (trial < 6) bad.resize(raw.size() * (size_t)(trial + 1) / 8);           // cut off
    else for (int k = 0; k < 1 + trial; ++k) bad[rng() % bad.size()] ^= (char)(1u << (rng() % 8));   // bit flips
    std::ofstream(tmp, std::ios::binary).write(bad.data(), (std::streamsize)bad.size());
    walk(tmp, false);
  }
  remove(tmp.c_str());
}

// IntSpan's sparse form (the status array inside the marked runs only, as the pipeline hands it to the block tests) against the
// dense array it stands for: every index inside a range, at its ends, between ranges and outside all of them.
static void span_checks() {
  std::vector<int> dense(5000, 0), values;
  std::vector<rsih::IntSpan::Range> ranges;
  const int bounds[][2] = {{0, 0}, {7, 19}, {20, 20}, {100, 1099}, {4990, 4999}};
  int v = 1;
  for (const auto& b : bounds) {
    ranges.push_back({b[0], b[1], (int64_t)values.size()});
    for (int i = b[0]; i <= b[1]; ++i) { dense[(size_t)i] = (v % 7) - 3 ? (v % 7) - 3 : 5; values.push_back(dense[(size_t)i]); ++v; }
  }
  const rsih::IntSpan sparse(values.data(), (int64_t)dense.size(), &ranges), full(dense);
  for (int64_t i = 0; i < (int64_t)dense.size(); ++i) CHECK(sparse[i] == full[i], "sparse span differs from the dense array");
  CHECK(sparse.at(100)[999] == dense[1099] && *sparse.at(4999) == dense[4999] && *sparse.at(50) == 0, "sparse span: at()");
}

// rsi_hot_run's narrowed upload (host_calls.cpp: narrow_depth_u8): bytes + list give the int32 array back, value for value --
// depths of 254 / 255 / 256, 32767 / 32768 / 65535 / 65536 (a saturating pack reads 16-bit intermediates as SIGNED: the first
// version turned everything from 32768 on into 0), INT32_MAX, negative ones; every length around the 32-value vector loop; a list
// that is too short reports how many there were.
static void narrow_checks() {
  uint64_t rs = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
  const int32_t odd[] = {254, 255, 256, 300, 32767, 32768, 65535, 65536, 70000, 1 << 24, 2147483647, -1, -2147483647 - 1, 0};
  for (int64_t n : {0, 1, 31, 32, 33, 63, 64, 65, 1000, 700001}) {
    std::vector<int32_t> d((size_t)n);
    for (auto& x : d) x = (int32_t)(rnd() % 90);
    for (int k = 0; k < 14 && n > 0; ++k) d[(size_t)(rnd() % (uint64_t)n)] = odd[k];
    for (int k = 0; k < 200 && n > 1000; ++k) d[(size_t)(rnd() % (uint64_t)n)] = (int32_t)(255 + rnd() % 3000000);
    const int64_t cap = n / 64 + 16;
    std::vector<uint8_t> b((size_t)n + 64, 0xAB);
    std::vector<int32_t> pos((size_t)cap), val((size_t)cap);
    const int64_t ne = rsih::narrow_depth_u8(d.data(), n, b.data(), pos.data(), val.data(), cap);
    int64_t expect = 0;
    for (int32_t x : d) expect += (uint32_t)x >= 255u;
    CHECK(ne == expect, "narrow_depth_u8: %lld escapes reported, %lld present (n = %lld)", (long long)ne, (long long)expect, (long long)n);
    if (ne <= cap) {
      std::vector<int32_t> back((size_t)n);
      for (int64_t i = 0; i < n; ++i) back[(size_t)i] = b[(size_t)i];
      for (int64_t k = 0; k < ne; ++k) { CHECK(pos[(size_t)k] >= 0 && pos[(size_t)k] < n && b[(size_t)pos[(size_t)k]] == 255, "narrow_depth_u8: bad list entry"); back[(size_t)pos[(size_t)k]] = val[(size_t)k]; }
      CHECK(back == d, "narrow_depth_u8: bytes + list do not give the array back (n = %lld)", (long long)n);
    }
    for (int k = 0; k < 64; ++k) CHECK(b[(size_t)n + k] == 0xAB, "narrow_depth_u8 wrote behind the array");
  }
  std::vector<int32_t> deep(4096, 1000), pos(8), val(8);
  std::vector<uint8_t> b(4096 + 64);
  CHECK(rsih::narrow_depth_u8(deep.data(), 4096, b.data(), pos.data(), val.data(), 8) == 4096, "narrow_depth_u8: a list that is too short must report the full count");
}

int main(int argc, char** argv) {
  quantile_checks();
  span_checks();
  narrow_checks();
  orc_params P;
  orc_default_params(&P);
  candidate_stage_case(0x5A11, 400007, 0, P);
  orc_params Q = P; Q.m = 51; Q.trans = 1;
  candidate_stage_case(0x5A12, 350013, 1, Q);
  orc_params R = P; R.merge = 0; R.chklen = 1.5; R.maxchkbp = 2000;
  candidate_stage_case(0x5A13, 300000, 1, R);
  candidate_stage_case(0x5A14, 2000003, 1, P, 1);
  if (argc > 1) bam_checks(argv[1]);
  if (g_fail) { fprintf(stderr, "%d check(s) failed\n", g_fail); return 1; }
  printf("host harness ok\n");
  return 0;
}
