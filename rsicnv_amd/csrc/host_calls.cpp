// host_calls.cpp -- candidate tests, boundary refinement, merge and final filters on the host
// (SURVEY.md 8a rows A15-A19).  Reference lines are cited per function (paths relative to
// /root/reference/src).  Compiled with -ffp-contract=off: the statistics below decide which
// candidates survive, so they have to round like the reference's x86-64 build.
#include "host_calls.h"
#include <immintrin.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include "hostmath.h"

namespace rsih {

static inline double tick_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------------------------------
namespace {
inline void bucket_indices_body(const float* x, size_t n, double lo, double dy, uint32_t* idx) {
  for (size_t i = 0; i < n; ++i) idx[i] = (uint32_t)(int64_t)(((double)x[i] - lo) / dy + 0.5);
}
__attribute__((target("avx2"))) void bucket_indices_avx2(const float* x, size_t n, double lo, double dy, uint32_t* idx) {
  bucket_indices_body(x, n, lo, dy, idx);
}
void bucket_indices_sse2(const float* x, size_t n, double lo, double dy, uint32_t* idx) { bucket_indices_body(x, n, lo, dy, idx); }
}  // namespace

void bucket_indices_f32(const float* x, size_t n, double lo, double dy, uint32_t* idx) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) bucket_indices_avx2(x, n, lo, dy, idx); else bucket_indices_sse2(x, n, lo, dy, idx);
}

// ---- depth from host memory, narrowed before it crosses PCIe (rsi_hot_run, pipeline.hip) ----
// dst[i] = src[i] for 0 <= src[i] < 255, else 255 ("look in the list"); the values that do not fit go to (esc_pos, esc_val), up to
// `cap` of them.  Returns their number (more than cap: the caller sends the int32 array as before).  The caller's array is what
// the reference's loaders leave in Array<int> RD (loaddata.cpp:519-531); at sequencing depths next to nothing escapes.
namespace {
inline int64_t narrow_scalar(const int32_t* src, int64_t i0, int64_t i1, uint8_t* dst, int32_t* esc_pos, int32_t* esc_val, int64_t cap, int64_t nesc) {
  for (int64_t i = i0; i < i1; ++i) {
    const int32_t v = src[i];
    if ((uint32_t)v < 255u) dst[i] = (uint8_t)v;
    else { dst[i] = 255; if (nesc < cap) { esc_pos[nesc] = (int32_t)i; esc_val[nesc] = v; } ++nesc; }
  }
  return nesc;
}
__attribute__((target("avx2"))) int64_t narrow_avx2(const int32_t* src, int64_t n, uint8_t* dst, int32_t* esc_pos, int32_t* esc_val, int64_t cap) {
  int64_t nesc = 0, i = 0;
  const __m256i order = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7);
  const __m256i ff = _mm256_set1_epi8((char)0xff);
  for (; i + 32 <= n; i += 32) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i)), b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 8));
    const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 16)), d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 24));
    // clamp at 255 first (the second pack reads its input as SIGNED 16 bits: 65535 would come out as 0), then 32 -> 16 -> 8; the packs
    // interleave the 128-bit halves: one permute puts the dwords back in order
    const __m256i lim = _mm256_set1_epi32(255);
    const __m256i bytes = _mm256_permutevar8x32_epi32(_mm256_packus_epi16(_mm256_packus_epi32(_mm256_min_epi32(a, lim), _mm256_min_epi32(b, lim)),
                                                                          _mm256_packus_epi32(_mm256_min_epi32(c, lim), _mm256_min_epi32(d, lim))), order);
    _mm256_storeu_si256(reinterpret_cast<__m256i*>(dst + i), bytes);
    const int neg = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_or_si256(_mm256_or_si256(a, b), _mm256_or_si256(c, d))));   // a negative depth saturates to 0
    if (neg | _mm256_movemask_epi8(_mm256_cmpeq_epi8(bytes, ff))) nesc = narrow_scalar(src, i, i + 32, dst, esc_pos, esc_val, cap, nesc);
  }
  return narrow_scalar(src, i, n, dst, esc_pos, esc_val, cap, nesc);
}
}  // namespace
int64_t narrow_depth_u8(const int32_t* src, int64_t n, uint8_t* dst, int32_t* esc_pos, int32_t* esc_val, int64_t cap) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  return avx2 ? narrow_avx2(src, n, dst, esc_pos, esc_val, cap) : narrow_scalar(src, 0, n, dst, esc_pos, esc_val, cap, 0);
}

double normal_cdf(double x) {
  // erf / erfc rational approximations of Cephes ndtr as used by alglib
  // (alglib/specialfunctions.cpp:3152-3302); numerators/denominators highest power first.
  static const double kErfNum[7] = {0.007547728033418631287834, -0.288805137207594084924010, 14.3383842191748205576712,
                                    38.0140318123903008244444, 3017.82788536507577809226, 7404.07142710151470082064,
                                    80437.3630960840172832162};
  static const double kErfDen[7] = {0.0, 1.00000000000000000000000, 38.0190713951939403753468, 658.070155459240506326937,
                                    6379.60017324428279487120, 34216.5257924628539769006, 80437.3630960840172826266};
  static const double kErfcNum[9] = {0.0, 0.5641877825507397413087057563, 9.675807882987265400604202961,
                                     77.08161730368428609781633646, 368.5196154710010637133875746,
                                     1143.262070703886173606073338, 2320.439590251635247384768711,
                                     2898.0293292167655611275846, 1826.3348842295112592168999};
  static const double kErfcDen[9] = {1.0, 17.14980943627607849376131193, 137.1255960500622202878443578,
                                     661.7361207107653469211984771, 2094.384367789539593790281779,
                                     4429.612803883682726711528526, 6089.5424232724435504633068,
                                     4958.82756472114071495438422, 1826.3348842295112595576438};
  auto horner = [](const double* c, int n, double t) { double a = c[0]; for (int i = 1; i < n; ++i) a = c[i] + t * a; return a; };
  const double z = x / 1.41421356237309504880;
  const double sign = z > 0 ? 1.0 : (z < 0 ? -1.0 : 0.0);
  const double az = fabs(z);
  double erf_z;
  if (az < 0.5) {
    const double t = az * az;
    erf_z = sign * 1.1283791670955125738961589031 * az * horner(kErfNum, 7, t) / horner(kErfDen, 7, t);
  } else if (az >= 10) {
    erf_z = sign;
  } else {
    const double erfc_az = exp(-(az * az)) * horner(kErfcNum, 9, az) / horner(kErfcDen, 9, az);
    erf_z = sign * (1 - erfc_az);
  }
  return 0.5 * (erf_z + 1);
}

// ------------------------------------------------------------------------------------------
DepthPager::DepthPager(const int32_t* d_ptr, int64_t n, hipStream_t stream, std::function<int32_t*()> mirror, void* staging, size_t staging_bytes,
                       hipEvent_t sync_event)
    : d_(d_ptr), n_(n), stream_(stream), mirror_source_(std::move(mirror)), have_((size_t)((n + (1 << kBits) - 1) >> kBits) + 1, 0),
      staging_(static_cast<int32_t*>(staging)), staging_elems_((int64_t)(staging_bytes / sizeof(int32_t))), sync_ev_(sync_event) {}

DepthPager::DepthPager(const int32_t* host_ptr, int64_t n)
    : d_(host_ptr), n_(n), stream_(nullptr), mirror_(const_cast<int32_t*>(host_ptr)),
      have_((size_t)((n + (1 << kBits) - 1) >> kBits) + 1, 1) {}

void DepthPager::wait() {
  if (!sync_ev_) { (void)hipStreamSynchronize(stream_); return; }
  (void)hipEventRecord(sync_ev_, stream_);
  for (int spin = 0; spin < 2000; ++spin) if (hipEventQuery(sync_ev_) != hipErrorNotReady) return;
  (void)hipEventSynchronize(sync_ev_);
}

void DepthPager::fetch(int64_t p0, int64_t p1) {
  const double t0 = tick_ms();
  if (!mirror_) {
    mirror_ = mirror_source_ ? mirror_source_() : nullptr;
    if (!mirror_) { failed_ = true; return; }   // the caller checks failed() and fails the run (RSI_ERR_INTERNAL)
  }
  const int64_t lo = p0 << kBits;
  int64_t hi = ((p1 + 1) << kBits);
  if (hi > n_) hi = n_;
  if (hi > lo) {
    if (staging_ && staging_elems_ > 0) {   // DMA into pinned memory, then a plain copy into the mirror
      for (int64_t a = lo; a < hi; a += staging_elems_) {
        const int64_t k = std::min(staging_elems_, hi - a);
        (void)hipMemcpyAsync(staging_, d_ + a, (size_t)k * sizeof(int32_t), hipMemcpyDeviceToHost, stream_);
        wait();
        memcpy(mirror_ + a, staging_, (size_t)k * sizeof(int32_t));
      }
    } else {
      (void)hipMemcpyAsync(mirror_ + lo, d_ + lo, (size_t)(hi - lo) * sizeof(int32_t), hipMemcpyDeviceToHost, stream_);
      wait();
    }
    fetched_ += (hi - lo) * (int64_t)sizeof(int32_t);
  }
  for (int64_t p = p0; p <= p1; ++p) have_[(size_t)p] = 1;
  fetch_ms_ += tick_ms() - t0;
}

void DepthPager::prefetch(int64_t lo, int64_t hi) {
  if (lo < 0) lo = 0;
  if (hi > n_ - 1) hi = n_ - 1;
  if (hi < lo) return;
  const int64_t p_lo = lo >> kBits, p_hi = hi >> kBits;
  for (int64_t p = p_lo; p <= p_hi;) {
    if (have_[(size_t)p]) { ++p; continue; }
    int64_t q = p;
    while (q + 1 <= p_hi && !have_[(size_t)(q + 1)]) ++q;
    fetch(p, q);
    p = q + 1;
  }
}

namespace {

struct VecView {   // bin-space arrays
  IntSpan v;
  int64_t size() const { return v.n; }
  int operator[](int64_t i) const { return v.p[i]; }
  void prefetch(int64_t, int64_t) const {}
  const int* raw() const { return v.p; }   // whole array
};
struct PagedView {
  DepthPager* p;
  int64_t size() const { return p->size(); }
  int operator[](int64_t i) const { return (*p)[i]; }
  void prefetch(int64_t lo, int64_t hi) const { p->prefetch(lo, hi); }
  const int* raw() const { return p->raw(); }    // valid only inside ranges that were prefetched
};

// ---- isitcnv's decision part (rsi.cpp:126-169) from the statistics of the two arrays ----
void finish_judgement(const CallerInput& in, const TestStats& st, Candidate& c) {
  double spread = sqrt(st.ref_var);
  if (spread < 1E-3) spread = st.ref_med / 40.0 + 1E-3;
  c.length = c.end - c.start + 1;
  c.cnvmed = st.cnv_med;
  c.cnvsd = sqrt(st.cnv_var);
  c.cnviqr = st.cnv_uqt - st.cnv_lqt;
  c.refmed = st.ref_med;
  c.refiqr = st.ref_uqt - st.ref_lqt;
  c.refsd = c.refiqr / 1.349;
  c.geno = 1;
  c.status = 1;
  const int observed = c.cnvmed > in.RDmedian ? kDup : kDel;
  if (c.type == kUnknown) c.type = observed;
  if (c.type != observed) { c.status = -9; return; }   // "basic assignment error"
  if (c.type == kDel) {
    double level = std::min(st.ref_med, in.RDmedian);
    level = std::max(level, 0.8 * in.RDmedian);
    const double nu = (3.0 * c.cnvmed - 2.0 * level) / spread;
    c.p1 = normal_cdf(nu);
    if (nu > 0) { c.status = -9; c.geno = 0; }
  } else {
    const double level = std::max(st.ref_med, in.RDmedian);
    const double nu = (2.5 * c.cnvmed - 3.0 * level) / spread / 1.5;
    c.p1 = 1.0 - normal_cdf(nu);
    if (nu < 0) { c.status = -9; c.geno = 0; }
  }
}

// ---- isitcnv (rsi.cpp:101-172): statistics of the candidate against its neighbourhood ----
void judge(const CallerInput& in, const std::vector<int>& ref, const std::vector<int>& body, Candidate& c) {
  const double t0 = tick_ms();
  const int width = (int)body.size();
  const int nwin = (int)ref.size() - width;
  if (nwin <= 0 || width <= 0) {   // the reference aborts here (CallerInput::short_neighbourhoods)
    if (in.short_neighbourhoods) ++*in.short_neighbourhoods;
    c.geno = 0; c.status = -9;
    return;
  }
  std::vector<float> winmean((size_t)nwin);
  // running mean of width `width` (rsi.cpp:113-124); its minimum, maximum, mean and second moment are
  // accumulated in the same index order as the reference's separate passes would
  double acc = 0;
  for (int i = 0; i < width; ++i) acc += ref[i];
  float flo = 0, fhi = 0;
  double msum = 0, s1 = 0, s2 = 0;
  for (int i = 0; i < nwin; ++i) {
    if (i > 0) acc = acc - ref[i - 1] + ref[i - 1 + width];
    const float w = (float)(acc / double(width));
    winmean[i] = w;
    if (i == 0) { flo = fhi = w; }
    flo = w < flo ? w : flo; fhi = w > fhi ? w : fhi;
    msum += w;                                   // partition_stat_tp's mean (wufunctions.cpp:372-378)
    s1 += (double)w; s2 += (double)w * (double)w;   // variancetp (wufunctions.cpp:790-796)
  }
  const double t1 = tick_ms();
  const Quantiles qr = grid_quantiles_f32_known(winmean.data(), winmean.size(), flo, fhi, msum);
  const double t2 = tick_ms();
  const double mu = s1 / double(nwin);
  double spread = sqrt(s2 / double(nwin) - mu * mu);
  if (spread < 1E-3) spread = qr.med / 40.0 + 1E-3;
  const double t3 = tick_ms();
  const Quantiles qc = grid_quantiles(body.data(), body.size());
  if (in.prof) { in.prof->winmean += t1 - t0; in.prof->quantiles += (t2 - t1) + (tick_ms() - t3); in.prof->variance += t3 - t2; in.prof->tests++; }
  TestStats st;
  st.cnv_lqt = qc.lqt; st.cnv_med = qc.med; st.cnv_uqt = qc.uqt; st.cnv_var = variance_pop(body.data(), body.size());
  st.ref_lqt = qr.lqt; st.ref_med = qr.med; st.ref_uqt = qr.uqt; st.ref_var = s2 / double(nwin) - mu * mu;
  (void)spread;
  finish_judgement(in, st, c);
}

// ---- isitcnvwrap (rsi.cpp:175-287): collect the reference neighbourhood around list[ci] ----
template <class View>
void test_candidate(const CallerInput& in, const View& A, std::vector<Candidate>& list, int ci) {
  const rsi_params& P = in.P;
  const int64_t N = A.size();
  const int count = (int)list.size();
  const Candidate me = list[ci];
  const int kind = me.type;
  const int body_len = me.end - me.start + 1;
  const int span_all = (int)in.ncompact;   // rsi::end - rsi::start + 1
  int d = body_len;
  if (N == span_all) { if (d < P.m * P.minmlen) d = (int)(P.m * P.minmlen); }
  if (N < span_all / 2) { if (d < P.minmlen) d = (int)P.minmlen + 1; }
  const double tg0 = tick_ms();
  const int capacity = (int)(P.chklen * d * 2);
  std::vector<int> ref((size_t)capacity, 0);
  const double too_high = in.RDmedian * 3.0, too_low = in.RDmedian * 0.15;
  auto skip_value = [&](int v) { return (kind == kDel && v > too_high) || (kind == kDup && v < too_low); };
  const int margin = int(body_len * P.buffer + 1);
  A.prefetch((int64_t)me.start - margin - 3LL * capacity / 5 - 64, (int64_t)me.end + margin + 3LL * capacity / 5 + 64);

  // left side, filled from slot `fill` downwards
  int pos = me.start - margin;
  int nb = ci - 1;
  while (pos > 0 && nb > 0 && pos < list[nb].start) --nb;
  while (nb > 0 && list[nb].status == -9) --nb;
  int fill = (int)(P.chklen * d - 1);
  if (N - me.end < P.chklen * d) fill = capacity - 1 - (int)N + me.end;
  const int top = fill;
  while (pos > 2 && fill >= 0) {
    --pos;
    const int v = A[pos];
    if (skip_value(v)) continue;
    if (nb >= 0 && pos >= list[nb].start && pos <= list[nb].end) {   // jump over another candidate
      pos = list[nb].start - 1;
      --nb;
      while (nb > 0 && list[nb].status == -9) --nb;
      continue;
    }
    ref[fill--] = v;
  }
  int used;
  if (fill >= 0) {   // ran out of sequence: close the gap at the front
    used = 0;
    for (int s = fill + 1; s <= top; ++s) ref[used++] = ref[s];
  } else {
    used = top + 1;
  }
  // right side, appended
  pos = me.end + margin;
  nb = ci + 1;
  while (pos < N - 2 && nb < count && pos > list[nb].end) ++nb;
  while (nb < count - 1 && list[nb].status == -9) ++nb;
  while (pos < N - 2 && used < 2 * P.chklen * d) {
    ++pos;
    const int v = A[pos];
    if (skip_value(v)) continue;
    if (nb < count && pos >= list[nb].start && pos <= list[nb].end) {
      pos = list[nb].end + 1;
      ++nb;
      while (nb < count - 1 && list[nb].status == -9) ++nb;
      continue;
    }
    if (used >= capacity) break;
    ref[used++] = v;
  }
  if (used < capacity) ref.resize((size_t)used);

  std::vector<int> body((size_t)body_len);
  for (int i = 0; i < body_len; ++i) body[i] = A[me.start + i];
  const int total = (int)ref.size() + (int)body.size();
  const int budget = P.maxchkbp * 10;
  if (total > budget) {   // thin both proportionally (rsi.cpp:264-282)
    const int nref = (int)((double)ref.size() / (double)total * (double)budget);
    const int nbody = (int)((double)body.size() / (double)total * (double)budget);
    std::vector<int> thin((size_t)nref);
    for (int i = 0; i < nref; ++i) thin[i] = ref[(size_t)(int)(double(i) / double(nref) * double(ref.size()))];
    ref.swap(thin);
    thin.assign((size_t)nbody, 0);
    for (int i = 0; i < nbody; ++i) thin[i] = body[(size_t)(int)(double(i) / double(nbody) * double(body.size()))];
    body.swap(thin);
  }
  if (in.prof) in.prof->gather += tick_ms() - tg0;
  judge(in, ref, body, list[ci]);
}

// ---- the list-only part of isitcnvwrap (rsi.cpp:175-257) for a device test of list[ci] on an array of N values ----
constexpr int kChainMax = 48;   // neighbour intervals handed to the device per side; it reports when it needs more
TestPlan plan_test(const CallerInput& in, int64_t N, const std::vector<Candidate>& list, int ci, int* cut) {
  const rsi_params& P = in.P;
  const int count = (int)list.size();
  const Candidate& me = list[ci];
  const int body_len = me.end - me.start + 1;
  const int span_all = (int)in.ncompact;
  int d = body_len;
  if (N == span_all) { if (d < P.m * P.minmlen) d = (int)(P.m * P.minmlen); }
  if (N < span_all / 2) { if (d < P.minmlen) d = (int)P.minmlen + 1; }
  TestPlan T;
  T.start = me.start; T.end = me.end; T.kind = me.type;
  T.capacity = (int)(P.chklen * d * 2);
  T.margin = int(body_len * P.buffer + 1);
  T.top = (int)(P.chklen * d - 1);
  if (N - me.end < P.chklen * d) T.top = T.capacity - 1 - (int)N + me.end;
  T.right_cap = 2 * P.chklen * d;
  T.budget = P.maxchkbp * 10;
  *cut = 0;
  // left: the neighbours the downward walk meets, in the order its `idx` pointer visits them
  int pos = me.start - T.margin;
  int nb = ci - 1;
  while (pos > 0 && nb > 0 && pos < list[nb].start) --nb;
  while (nb > 0 && list[nb].status == -9) --nb;
  while (nb >= 0 && (int)T.left_chain.size() < kChainMax) {
    T.left_chain.push_back({list[nb].start, list[nb].end});
    --nb;
    while (nb > 0 && list[nb].status == -9) --nb;
  }
  if (nb >= 0) *cut |= 1;
  // right
  pos = me.end + T.margin;
  nb = ci + 1;
  while (pos < N - 2 && nb < count && pos > list[nb].end) ++nb;
  while (nb < count - 1 && list[nb].status == -9) ++nb;
  while (nb < count && (int)T.right_chain.size() < kChainMax) {
    T.right_chain.push_back({list[nb].start, list[nb].end});
    ++nb;
    while (nb < count - 1 && list[nb].status == -9) ++nb;
  }
  if (nb < count) *cut |= 2;
  // bit 2: the left walk certainly fills its top + 1 slots unless every other base is extreme -- more than twice as many
  // positions between the chromosome's start and the candidate, the listed neighbours taken off, as slots.  The device's
  // right walk then gathers only what the left one leaves (kernels_cand.hip: k_cand_gather) and the result is checked there.
  if (!(*cut & 1) && T.top >= 0) {
    long long avail = (long long)me.start - T.margin - 2;
    for (const auto& iv : T.left_chain) avail -= (long long)iv.second - iv.first + 1;
    if (avail >= 2LL * (T.top + 1) + 64) *cut |= 4;
  }
  T.cut = *cut;
  return T;
}

bool same_plan(const TestPlan& a, const TestPlan& b) {
  return a.start == b.start && a.end == b.end && a.kind == b.kind && a.margin == b.margin && a.capacity == b.capacity &&
         a.top == b.top && a.budget == b.budget && a.right_cap == b.right_cap && a.cut == b.cut && a.left_chain == b.left_chain &&
         a.right_chain == b.right_chain;
}

// Results of tests launched ahead of the list logic that decides whether they are needed: a result is
// used only if the plan built from the list as it really is at that moment equals the guessed one.
struct Speculation {
  std::vector<TestPlan> plans;
  std::vector<TestStats> stats;
  std::vector<int> reach;
  std::vector<char> ok, have;
  void run(const CallerInput& in) {
    if (plans.empty()) return;
    if (!in.tester->test(plans, stats, reach, ok)) ok.assign(plans.size(), 0);
    have.assign(plans.size(), 1);
  }
};

// test of list[ci] against the per-base depth: on the device when a tester is attached (using a
// speculative result when its plan still holds), else -- or when the device declines -- on the host
void test_bases(const CallerInput& in, const PagedView& A, std::vector<Candidate>& list, int ci, const Speculation* spec = nullptr,
                int slot = -1) {
  if (!in.tester) { test_candidate(in, A, list, ci); return; }
  const double t0 = tick_ms();
  int cut = 0;
  TestPlan plan = plan_test(in, A.size(), list, ci, &cut);
  TestStats st{};
  bool ok = false;
  if (spec && slot >= 0 && slot < (int)spec->plans.size() && spec->have[slot] && same_plan(plan, spec->plans[slot])) {
    ok = spec->ok[slot] != 0;
    st = spec->stats[slot];
    if (in.prof) in.prof->spec_hits++;
  } else {
    std::vector<TestPlan> one(1, plan);
    std::vector<TestStats> sts;
    std::vector<int> reach;
    std::vector<char> oks;
    ok = in.tester->test(one, sts, reach, oks) && oks[0];
    if (ok) st = sts[0];
    if (in.prof) in.prof->single_tests++;
  }
  if (in.prof) { in.prof->device_ms += tick_ms() - t0; in.prof->tests++; }
  if (!ok) { if (in.prof) in.prof->host_fallbacks++; test_candidate(in, A, list, ci); return; }
  finish_judgement(in, st, list[ci]);
}

// ---- get_continuous_segments (rsi.cpp:291-327): runs of equal-sign marks, last run not emitted ----
void marked_runs(const std::vector<int>& marks, std::vector<Candidate>& out) {
  out.clear();
  bool open = false;
  int first = 0, last = 0;
  for (int i = 0; i < (int)marks.size(); ++i) {
    if (marks[i] == 0) continue;
    if (!open) { first = last = i; open = true; continue; }
    if ((double)marks[i] * (double)marks[last] > 0 && (i - last) <= 1) { last = i; continue; }
    Candidate c; c.start = first; c.end = last;
    out.push_back(c);
    first = last = i;
  }
}

// ---- multisegments (rsi.cpp:368-410): nested level sets of a rejected segment ----
void nested_levels(const Candidate& seg, IntSpan status, std::vector<Candidate>& out) {
  out.clear();
  const int len = seg.end - seg.start + 1;
  int lo = status[seg.start], hi = status[seg.start];
  for (int i = 0; i < len; ++i) { lo = std::min(lo, status[seg.start + i]); hi = std::max(hi, status[seg.start + i]); }
  std::vector<int> member((size_t)len);
  std::vector<Candidate> runs;
  for (int level = lo; level < hi; ++level) {
    if (level == 0) continue;
    bool present = false;
    for (int i = 0; i < len; ++i) {
      const int s = status[seg.start + i];
      member[i] = 0;
      if (s == 0) continue;
      if (s == level) present = true;
      if (level < 0 && s < 0 && s >= level) member[i] = 1;
      if (level > 0 && s > 0 && s <= level) member[i] = 1;
    }
    if (!present) continue;
    marked_runs(member, runs);
    for (Candidate& r : runs) {
      r.start = std::max(r.start + seg.start, seg.start);
      r.end = std::min(r.end + seg.start, seg.end);
      out.push_back(r);
    }
  }
}

// ---- sortcnvstartposition (rsi.cpp:549-577): stable by start (arrayindex_tp is a stable gnome sort) ----
void order_by_start(std::vector<Candidate>& L) {
  for (Candidate& c : L) if (c.start > c.end) std::swap(c.start, c.end);
  std::stable_sort(L.begin(), L.end(), [](const Candidate& a, const Candidate& b) { return a.start < b.start; });
}

// ---- optimize_with_derivative (rsi.cpp:889-944) ----
// The reference builds the whole vector dd[i] = (sum of the len values left of from+i) - (sum of
// the len values from from+i on) and then looks for its first maximum / minimum (DEL) or minimum /
// maximum (DUP) over the first and the last 2*reach entries.  The sums are integers below 2^53, so
// they are tracked exactly in int64 and both searches run in the same sweep without storing dd.
template <class View>
void sharpen_edges(const View& A, Candidate& c) {
  const int len = c.end - c.start + 1;
  const int reach = std::max(250, len / 4);
  const int from = c.start - reach, to = c.end + reach;
  if (from < 2 * len) return;
  if (to > A.size() - 2 * len) return;
  A.prefetch((int64_t)from - len - 1, (int64_t)to + len + 1);
  const int* p = A.raw();
  if (!p) return;   // the pager has no host memory for its mirror (DepthPager::failed(): the caller fails the run)
  int64_t diff = 0;
  for (int k = from - len; k < from; ++k) diff += p[k];
  for (int k = from; k < from + len; ++k) diff -= p[k];
  const int nstep = to - from;                 // dd.size()
  const int tail0 = nstep - 2 * reach;         // first index of the second search
  const bool del = c.type == kDel, dup = c.type == kDup;
  int best_lo = -1, best_hi = -1;
  int64_t ext_lo = 0, ext_hi = 0;
  for (int i = 0; i < nstep; ++i) {
    if (i > 0) { const int q = from + i; diff = diff - p[q - 1 - len] + 2 * (int64_t)p[q - 1] - p[q - 1 + len]; }
    if (i < 2 * reach) {
      if (del && diff > ext_lo) { ext_lo = diff; best_lo = i; }
      if (dup && diff < ext_lo) { ext_lo = diff; best_lo = i; }
    }
    if (i >= tail0) {
      if (del && diff < ext_hi) { ext_hi = diff; best_hi = i; }
      if (dup && diff > ext_hi) { ext_hi = diff; best_hi = i; }
    }
  }
  if (best_lo > 0) c.start = from + best_lo;
  if (best_hi > 0) c.end = to - nstep + best_hi;
}

template <class View>
double range_mean(const View& A, int lo, int hi) {   // mean_tp, wufunctions.cpp:666-690
  A.prefetch(lo, hi);
  double s = 0;
  for (int i = lo; i <= hi; ++i) s += (double)A[i];
  return s / double(hi - lo + 1);
}

void drop_deleted(std::vector<Candidate>& L) {
  std::vector<Candidate> keep;
  for (const Candidate& c : L) if (c.status != -9) keep.push_back(c);
  L.swap(keep);
}

// ---- mergesegments (rsi.cpp:694-885) ----
void merge_neighbours(const CallerInput& in, const PagedView& A, std::vector<Candidate>& L) {
  const rsi_params& P = in.P;
  std::vector<Candidate> T;
  // overlapping neighbours of one type
  for (int i = 0; i + 1 < (int)L.size(); ++i) {
    if (L[i].type != L[i + 1].type) continue;
    if (!(std::max(L[i].start, L[i + 1].start) < std::min(L[i].end, L[i + 1].end))) continue;
    Candidate joined = L[i];
    joined.start = std::min(L[i].start, L[i + 1].start);
    joined.end = std::max(L[i].end, L[i + 1].end);
    T = L; T[i] = joined; T[i + 1] = joined; T[i + 1].status = -9;
    test_bases(in, A, T, i);
    if (T[i].geno == 0) {   // the union fails: test each on its own
      T = L; T[i + 1].status = -9;
      test_bases(in, A, T, i);
      T[i].status = -9; T[i + 1].status = 0;
      test_bases(in, A, T, i + 1);
      if (T[i + 1].p1 < T[i].p1) T[i] = T[i + 1];
      if (T[i].p1 > P.p) { L[i].status = -9; L[i + 1].status = -9; }
    }
    if (T[i].geno == 0) continue;
    L[i] = T[i]; L[i].status = -9;
    L[i + 1] = T[i]; L[i + 1].status = 0;
  }
  drop_deleted(L);
  if (!P.merge) return;
  // nearby neighbours of one type
  auto near_pair = [&](int i) {
    if (L[i].type != L[i + 1].type) return false;
    if (L[i].geno == 0 || L[i + 1].geno == 0) return false;
    const int gap = L[i + 1].start - L[i].end;
    const int w1 = L[i].end - L[i].start, w2 = L[i + 1].end - L[i + 1].start;
    return !(gap > w1 * P.chklen * 0.7 && gap > w2 * P.chklen * 0.7);
  };
  // With a device tester the three range sums of every pair that qualifies on the list as it stands
  // are taken in one launch, and the joined candidates that pass the depth rule are tested in a
  // second one; a pair whose members changed through an earlier merge is redone on its own.
  const int npairs = std::max(0, (int)L.size() - 1);
  std::vector<int64_t> sums;
  std::vector<int> sum_slot((size_t)npairs, -1);
  std::vector<std::pair<int, int>> ranges;
  if (in.tester) {
    for (int i = 0; i < npairs; ++i) {
      if (!near_pair(i)) continue;
      sum_slot[i] = (int)ranges.size();
      ranges.push_back({L[i].start, L[i].end});
      ranges.push_back({L[i + 1].start, L[i + 1].end});
      ranges.push_back({L[i].start, L[i + 1].end});
    }
    if (!ranges.empty() && !in.tester->range_sums(ranges, sums)) { sums.clear(); std::fill(sum_slot.begin(), sum_slot.end(), -1); }
  }
  auto mean_of = [&](int slot, int lo, int hi) -> double {   // mean_tp, wufunctions.cpp:666-690 (integer sums are exact in double)
    if (slot >= 0 && ranges[slot].first == lo && ranges[slot].second == hi) return (double)sums[slot] / double(hi - lo + 1);
    if (in.tester) {
      std::vector<std::pair<int, int>> one(1, {lo, hi});
      std::vector<int64_t> s1;
      if (in.tester->range_sums(one, s1)) return (double)s1[0] / double(hi - lo + 1);
    }
    return range_mean(A, lo, hi);
  };
  auto depth_rule = [&](int i, int slot) {   // true: the pair may be joined (rsi.cpp:775-790)
    const int w1 = L[i].end - L[i].start, w2 = L[i + 1].end - L[i + 1].start;
    const double m1 = mean_of(slot, L[i].start, L[i].end), m2 = mean_of(slot < 0 ? -1 : slot + 1, L[i + 1].start, L[i + 1].end);
    const double both = (m1 * w1 + m2 * w2) / (w1 + w2);
    const double across = mean_of(slot < 0 ? -1 : slot + 2, L[i].start, L[i + 1].end);
    if (L[i].type == kDel && across > both + 1.5 * L[i + 1].refsd + 1.5 * L[i].refsd) return false;
    if (L[i].type == kDup && across < both - 1.5 * L[i + 1].refsd - 1.5 * L[i].refsd) return false;
    return true;
  };
  auto joined_list = [&](int i) {
    Candidate joined = L[i];
    joined.end = L[i + 1].end;
    T = L; T[i] = joined; T[i + 1] = joined; T[i + 1].status = -9;
  };
  Speculation spec;
  std::vector<int> spec_slot((size_t)npairs, -1);
  if (in.tester) {
    for (int i = 0; i < npairs; ++i) {
      if (sum_slot[i] < 0 || !depth_rule(i, sum_slot[i])) continue;
      joined_list(i);
      int cut = 0;
      spec_slot[i] = (int)spec.plans.size();
      spec.plans.push_back(plan_test(in, A.size(), T, i, &cut));
    }
    spec.run(in);
  }
  for (int i = 0; i + 1 < (int)L.size(); ++i) {
    if (!near_pair(i)) continue;
    if (!depth_rule(i, in.tester ? sum_slot[i] : -1)) continue;
    joined_list(i);
    test_bases(in, A, T, i, &spec, spec_slot[i]);
    if (T[i].geno == 0) continue;
    L[i] = T[i]; L[i + 1] = T[i]; L[i].status = -9;
  }
  drop_deleted(L);
}

// ---- expand_coordinate (rsi.cpp:1524-1551): compacted index -> reference index ----
int to_reference(const std::vector<Region>& noncode, int p) {
  int removed = 0;
  for (const Region& r : noncode) {
    const int grown = removed + (r.end - r.start + 1);
    if (p < r.end + 1 - grown) break;
    removed = grown;
  }
  return p + removed;
}

}  // namespace

// ---- areblockscnv (rsi.cpp:415-546) ----
void test_block_segments(const CallerInput& in, IntSpan status, std::vector<Candidate>& segs) {
  const VecView bins{in.binmedint};
  std::vector<Candidate> T = segs;
  // First round (rsi.cpp:430-436): test i sees which of the segments before it were just rejected (a rejected neighbour is no
  // longer jumped over).  With a device tester and enough segments to pay for its four launches, all of them are tested in one
  // batch against the list as it stands.  Such a result holds for test i unless a segment rejected since lies where i's LEFT
  // walk went (the segments behind i are untouched when i's turn comes; its own fields do not change): every rejected
  // segment must end before the last position that walk examined.  Comparing the whole plans, as the final tests do, is too
  // strict here -- most block segments are rejected, and each rejection changes the 48-entry neighbour chains of all tests
  // behind it, although the segments lie tens of thousands of bins apart and a walk covers a few hundred.  Anything else -- few
  // segments, a rejected neighbour within reach, a test the device declined -- is the host loop (14 us per test: 0.5 ms of a
  // 250 Mb chromosome's 3 ms).
  constexpr int kBlockBatchMin = 24;
  std::vector<TestPlan> plans;
  std::vector<TestStats> stats;
  std::vector<int> reach;
  std::vector<char> ok;
  bool sorted = true;
  for (size_t i = 1; i < T.size(); ++i) sorted = sorted && T[i - 1].end < T[i].start;
  for (const Candidate& c : T) sorted = sorted && c.status != -9;   // (nothing rejected before the round starts)
  if (in.block_tester && sorted && (int)T.size() >= kBlockBatchMin) {
    int cut = 0;
    for (int i = 0; i < (int)T.size(); ++i) plans.push_back(plan_test(in, bins.size(), T, i, &cut));
    if (!in.block_tester->test(plans, stats, reach, ok)) plans.clear();
  }
  int rejected_end = -1;   // the largest end among the segments rejected so far
  for (int i = 0; i < (int)T.size(); ++i) {
    if (!plans.empty() && ok[(size_t)i] && !(plans[(size_t)i].cut & 1) && rejected_end < reach[(size_t)i]) {
      finish_judgement(in, stats[(size_t)i], T[i]);
      if (in.prof) in.prof->block_batch_hits++;
    } else {
      test_candidate(in, bins, T, i);
    }
    if (T[i].status == -9 && T[i].end > rejected_end) rejected_end = T[i].end;
  }
  for (int i = 0; i < (int)T.size(); ++i) {
    if (T[i].status != -9) continue;
    if (T[i].type == kDel && T[i].cnvmed < 0.7 * T[i].refmed) { T[i].geno = 1; T[i].p1 = in.P.p; continue; }
    if (T[i].type == kDup && T[i].cnvmed > 1.3 * T[i].refmed) { T[i].geno = 1; T[i].p1 = in.P.p; continue; }
    const Candidate original = T[i];
    Candidate chosen = T[i];
    std::vector<Candidate> levels;
    nested_levels(chosen, status, levels);
    for (int j = (int)levels.size() - 1; j >= 0; --j) {
      levels[j].type = chosen.type;
      T[i] = levels[j];
      test_candidate(in, bins, T, i);
      levels[j] = T[i];
    }
    for (int j = (int)levels.size() - 1; j >= 0; --j) {
      if (levels[j].geno == 0) continue;
      if (chosen.geno == 0) chosen = levels[j];
      if (levels[j].length > chosen.length) chosen = levels[j];
    }
    if (chosen.geno == 0) chosen = original;
    T[i] = chosen;
  }
  segs.swap(T);
}

// ---- detectcnv after the block tests (rsi.cpp:1860-1931), then sd_filters (rsi.cpp:1753-1792) ----
void call_from_segments(const CallerInput& in, std::vector<Candidate> segs, DepthPager& depth,
                        std::vector<Candidate>& blocks, std::vector<Candidate>& raw, std::vector<Candidate>& kept) {
  const rsi_params& P = in.P;
  const PagedView bases{&depth};
  const int m = P.m, np = (int)in.ncompact;
  order_by_start(segs);
  blocks = segs;
  std::vector<Candidate> L;
  for (Candidate c : segs) {
    if (c.geno == 0 || c.start == c.end) continue;
    c.start = c.start * m + m / 2;   // bins -> bases (rsi.cpp:1868-1872)
    c.end = c.end * m + m / 2;
    if (c.start < 0) c.start = 0;
    if (c.end > np - 1) c.end = np - 1;
    c.length = c.end - c.start + 1;
    L.push_back(c);
  }
  const double ts0 = tick_ms();
  if (!(in.tester && in.tester->sharpen(L)))
    for (int pass = 0; pass < 2; ++pass) for (Candidate& c : L) sharpen_edges(bases, c);
  order_by_start(L);
  const double ts1 = tick_ms();
  merge_neighbours(in, bases, L);
  const double ts2 = tick_ms();
  order_by_start(L);
  // final tests (rsi.cpp:1893-1913).  Test i sees which earlier candidates were just rejected, so with
  // a device tester all tests are first run against the list as it stands, and a result is replaced
  // only where a rejection changed the plan of a later test.
  Speculation spec;
  if (in.tester) {
    int cut = 0;
    for (int i = 0; i < (int)L.size(); ++i) spec.plans.push_back(plan_test(in, bases.size(), L, i, &cut));
    spec.run(in);
  }
  raw.clear();
  for (int i = 0; i < (int)L.size(); ++i) {
    const double nbins = double(L[i].end - L[i].start + 1) / double(m);
    test_bases(in, bases, L, i, &spec, i);
    L[i].score = (L[i].cnvmed - in.RDmedian) * sqrt(nbins);
    const int r1 = to_reference(*in.noncode, L[i].start), r2 = to_reference(*in.noncode, L[i].end);
    for (const Region& g : *in.noncode) if (std::max(r1, g.start) <= std::min(r2, g.end)) L[i].status = -9;
    if (L[i].status != -9) raw.push_back(L[i]);
  }
  if (in.prof) { in.prof->sharpen += ts1 - ts0; in.prof->merge += ts2 - ts1; in.prof->final_tests += tick_ms() - ts2; }
  for (Candidate& c : raw) { c.start = to_reference(*in.noncode, c.start); c.end = to_reference(*in.noncode, c.end); }

  kept.clear();
  const int minlen = std::max(m * 2, 500);
  const double sd = in.RDsd / 1.2;
  for (const Candidate& c : raw) {
    const int span = abs(c.end - c.start);
    bool keep = !(span < 1000);
    if (c.type == kDel) {
      if (c.p1 > 0.2 || c.refsd > 0.6 * sd || c.cnvsd > 1.3 * sd) keep = false;
      if (c.cnvsd * in.RDmedian > 2.5 * c.cnvmed * sd) keep = false;
      if (c.cnvmed < 0.66 * std::min(in.RDmedian, c.refmed) && c.cnvsd < sd && span > 800) keep = true;
    }
    if (c.type == kDup) {
      if (c.p1 > 0.05 || c.refsd > 0.6 * sd) keep = false;
      if (c.cnvsd * in.RDmedian > 2.0 * c.cnvmed * sd) keep = false;
    }
    if (span < minlen) keep = false;
    if (keep) kept.push_back(c);
  }
}

}  // namespace rsih
