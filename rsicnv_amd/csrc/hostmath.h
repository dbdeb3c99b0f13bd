// hostmath.h -- host-side numeric helpers of the product path (small arrays only: candidate
// neighbourhoods, 31 MAD values, histogram walks of device-built histograms).
// The reference semantics they follow are cited per function; paths relative to
// /root/reference/src.  Compiled with -ffp-contract=off.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace rsih {

// Quantile triple of partition_stat_tp (wufunctions.cpp:364-424): lower quartile, "median", upper
// quartile on a grid of step dy anchored at the minimum; (min, mean, max) when max-min < dy.
struct Quantiles { double lqt, med, uqt; };

template <class T>
inline double grid_step(const T*) { return 0.01; }
template <>
inline double grid_step<int>(const int*) { return 1.0; }

// bucket index (size_t)((x - lo)/dy + 0.5) for a float array, written as uint32 (wufunctions.cpp:396).
// Separate from the counting loop so that the divisions vectorise; the AVX2 clone is selected at
// run time.  Same IEEE operations in every variant (no FMA, no reciprocal tricks).
void bucket_indices_f32(const float* x, size_t n, double lo, double dy, uint32_t* idx);

template <class T>
Quantiles grid_quantiles(const T* x, size_t n) {
  const double dy = grid_step(x);
  double lo = x[0], hi = x[0], acc = 0;
  for (size_t i = 0; i < n; ++i) {
    acc += x[i];
    if (x[i] < lo) lo = x[i];
    if (x[i] > hi) hi = x[i];
  }
  Quantiles q{lo, acc / (double)n, hi};
  if ((hi - lo) < dy) return q;
  const size_t buckets = (size_t)((hi - lo) / dy + 2);
  std::vector<uint32_t> cnt(buckets + 1, 0);
  for (size_t i = 0; i < n; ++i) cnt[(size_t)((x[i] - lo) / dy + 0.5)]++;
  const size_t r1 = n / 4, r2 = n / 2, r3 = n * 3 / 4;
  size_t seen = 0;
  for (size_t b = 0; b < buckets; ++b) {
    const size_t upto = seen + cnt[b];
    if (seen < r1 && upto >= r1) q.lqt = lo + b * dy;
    if (seen < r2 && upto >= r2) q.med = lo + b * dy;
    if (seen < r3 && upto >= r3) q.uqt = lo + b * dy;
    seen = upto;
  }
  return q;
}

// Specialisations for the two hot element types of the candidate tests.
// int: dy = 1, so (x - lo)/1 + 0.5 truncates to x - lo exactly -- no floating point per element.
template <>
inline Quantiles grid_quantiles<int>(const int* x, size_t n) {
  int lo = x[0], hi = x[0];
  double acc = 0;   // sums of ints are exact in double (n * max < 2^53)
  long long isum = 0;
  for (size_t i = 0; i < n; ++i) { isum += x[i]; lo = x[i] < lo ? x[i] : lo; hi = x[i] > hi ? x[i] : hi; }
  acc = (double)isum;
  Quantiles q{(double)lo, acc / (double)n, (double)hi};
  if (((double)hi - (double)lo) < 1.0) return q;
  const size_t buckets = (size_t)(((double)hi - (double)lo) / 1.0 + 2);
  std::vector<uint32_t> cnt(buckets + 1, 0);
  for (size_t i = 0; i < n; ++i) cnt[(size_t)((long long)x[i] - lo)]++;
  const size_t r1 = n / 4, r2 = n / 2, r3 = n * 3 / 4;
  size_t seen = 0;
  for (size_t b = 0; b < buckets; ++b) {
    const size_t upto = seen + cnt[b];
    if (seen < r1 && upto >= r1) q.lqt = (double)lo + b * 1.0;
    if (seen < r2 && upto >= r2) q.med = (double)lo + b * 1.0;
    if (seen < r3 && upto >= r3) q.uqt = (double)lo + b * 1.0;
    seen = upto;
  }
  return q;
}
// float, with minimum / maximum / running sum already known (accumulated in index order by the
// caller, double += float as the reference does).
inline Quantiles grid_quantiles_f32_known(const float* x, size_t n, float flo, float fhi, double sum) {
  const double dy = 0.01;
  const double lo = flo, hi = fhi;
  Quantiles q{lo, sum / (double)n, hi};
  if (n == 0 || (hi - lo) < dy) return q;
  const size_t buckets = (size_t)((hi - lo) / dy + 2);
  std::vector<uint32_t> cnt(buckets + 1, 0), idx(n);
  bucket_indices_f32(x, n, lo, dy, idx.data());
  for (size_t i = 0; i < n; ++i) cnt[idx[i]]++;
  const size_t r1 = n / 4, r2 = n / 2, r3 = n * 3 / 4;
  size_t seen = 0;
  for (size_t b = 0; b < buckets; ++b) {
    const size_t upto = seen + cnt[b];
    if (seen < r1 && upto >= r1) q.lqt = lo + b * dy;
    if (seen < r2 && upto >= r2) q.med = lo + b * dy;
    if (seen < r3 && upto >= r3) q.uqt = lo + b * dy;
    seen = upto;
  }
  return q;
}
template <>
inline Quantiles grid_quantiles<float>(const float* x, size_t n) {
  float flo = x[0], fhi = x[0];
  double acc = 0;
  for (size_t i = 0; i < n; ++i) { acc += x[i]; flo = x[i] < flo ? x[i] : flo; fhi = x[i] > fhi ? x[i] : fhi; }
  return grid_quantiles_f32_known(x, n, flo, fhi, acc);
}

// The same walk over an integer histogram built on the device: hist[v] = multiplicity of value v,
// v in [0, nvals); `total` values in all.  Returns false when the histogram is empty.
// (dy = 1: bucket index = v - min exactly, wufunctions.cpp:396 with integer data.)
inline bool hist_quantiles_int(const uint64_t* hist, size_t nvals, uint64_t total, Quantiles& q) {
  if (total == 0) return false;
  size_t lo = 0, hi = nvals - 1;
  while (lo < nvals && hist[lo] == 0) ++lo;
  while (hi > lo && hist[hi] == 0) --hi;
  if (lo >= nvals) return false;
  // the mean is only returned in the degenerate case, where every value equals lo
  q = Quantiles{(double)lo, (double)lo, (double)hi};
  if ((double)hi - (double)lo < 1.0) return true;
  const uint64_t r1 = total / 4, r2 = total / 2, r3 = total * 3 / 4;
  uint64_t seen = 0;
  for (size_t v = lo; v <= hi; ++v) {
    const uint64_t upto = seen + hist[v];
    if (seen < r1 && upto >= r1) q.lqt = (double)lo + (double)(v - lo) * 1.0;
    if (seen < r2 && upto >= r2) q.med = (double)lo + (double)(v - lo) * 1.0;
    if (seen < r3 && upto >= r3) q.uqt = (double)lo + (double)(v - lo) * 1.0;
    seen = upto;
  }
  return true;
}

// Median of a 0.01-grid histogram built on the device with anchor ymin (bucket b <-> ymin + b*0.01).
inline double hist_median_grid(const uint32_t* hist, size_t np, uint64_t total, double ymin) {
  const uint64_t r2 = total / 2;
  uint64_t seen = 0;
  double med = ymin;
  for (size_t b = 0; b < np; ++b) {
    const uint64_t upto = seen + hist[b];
    if (seen < r2 && upto >= r2) med = ymin + b * 0.01;
    seen = upto;
  }
  return med;
}

// variancetp with end_rule -1 (wufunctions.cpp:766-809)
template <class T>
double variance_pop(const T* y, size_t n) {
  double s1 = 0.0, s2 = 0.0;
  for (size_t i = 0; i < n; ++i) { s1 += (double)y[i]; s2 += (double)y[i] * (double)y[i]; }
  const double mu = s1 / double(n);
  return s2 / double(n) - mu * mu;
}

// Standard normal CDF as alglib::normaldistribution computes it (alglib/specialfunctions.cpp:
// 3152-3302, Cephes ndtr): rational approximations evaluated by Horner's rule.
double normal_cdf(double x);

// Largest non-negative double s for which pred(s) holds, pred being true on [0, s*] and false
// above (monotone).  Returns -1.0 when pred(0) is false and +inf when it never turns false.
template <class Pred>
double last_true(Pred pred) {
  if (!pred(0.0)) return -1.0;
  union { double d; uint64_t u; } lo, hi, mid;
  lo.d = 0.0; hi.d = INFINITY;
  if (pred(hi.d)) return INFINITY;
  while (hi.u - lo.u > 1) {   // non-negative doubles order like their bit patterns
    mid.u = lo.u + (hi.u - lo.u) / 2;
    if (pred(mid.d)) lo = mid; else hi = mid;
  }
  return lo.d;
}
// Smallest non-negative double s for which pred(s) holds, pred false below and true from s* on.
template <class Pred>
double first_true(Pred pred) {
  if (pred(0.0)) return 0.0;
  union { double d; uint64_t u; } lo, hi, mid;
  lo.d = 0.0; hi.d = INFINITY;
  if (!pred(hi.d)) return INFINITY;   // never: callers treat +inf as "no hit" (sum >= inf is false for finite sums)
  while (hi.u - lo.u > 1) {
    mid.u = lo.u + (hi.u - lo.u) / 2;
    if (pred(mid.d)) hi = mid; else lo = mid;
  }
  return hi.d;
}

}  // namespace rsih
