// oracle/rsi_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's read-depth CNV hot path (yhwu/rsicnv), written from the
// semantics of the reference, stage by stage.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this; the product (rsicnv_amd/) never does.
//
// Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every stage of this file against the
// real reference compiled into oracle/_ref/libref.so (oracle/Makefile.ref) on seeded inputs, and
// tests/golden/ holds outputs of that reference for the GPU box, where /root/reference is absent.
//
// Each function names the reference lines it follows (paths relative to /root/reference/src).
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (x86-64 SSE2 doubles, no FMA, no fast-math: the
// floating-point environment of the reference's own build, src/Makefile:6).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace orc {

enum { T_DEL = 0, T_DUP = 1, T_UNKNOWN = 2 };   // rsi.h:4-6

struct Params {          // rsi.cpp:34-98 (the flags the path reads)
  int32_t m, gcadjust, trans, merge, maxchkbp, debug;
  double cap, epsilon, threshold, chklen, minmlen, buffer, p;
};

struct Cnv {             // rsi.h:8-51
  int tid = -1, type = T_UNKNOWN, geno = 0, status = 0, start = 0, end = 0, length = 0;
  double score = 0, p1 = 1.0, p2 = 1.0, cnvmed = 0, cnvsd = 0, cnviqr = 0, refmed = 0, refsd = 0, refiqr = 0;
};

struct Region { int start, end; };   // inclusive, 0-based (rsi::noncodelist)

// ---------------------------------------------------------------------------------------------
// Histogram quantiles: partition_stat_tp, wufunctions.cpp:364-424 (macros wu2.h:5-8).
// Returns lower quartile / "median" / upper quartile on the dy grid; (min, mean, max) when
// max-min < dy.
template <class T>
void pstat(const T* x, size_t n, double dy, double& lqt, double& med, double& uqt) {
  double ymin = x[0], ymax = x[0], mean = 0;
  for (size_t i = 0; i < n; ++i) {
    mean += x[i];
    if (x[i] < ymin) ymin = x[i];
    if (x[i] > ymax) ymax = x[i];
  }
  mean /= (double)n;
  lqt = ymin; med = mean; uqt = ymax;
  if ((ymax - ymin) < dy) return;
  size_t np = (size_t)((ymax - ymin) / dy + 2);
  std::vector<size_t> h(np + 1, 0);
  for (size_t i = 0; i < n; ++i) {
    double idx = (x[i] - ymin) / dy + 0.5;
    h[(size_t)idx] += 1;
  }
  size_t c = 0, k1 = n / 4, k2 = n / 2, k3 = n * 3 / 4;
  for (size_t i = 0; i < np; ++i) {
    if (c < k1 && c + h[i] >= k1) lqt = ymin + i * dy;
    if (c < k2 && c + h[i] >= k2) med = ymin + i * dy;
    if (c < k3 && c + h[i] >= k3) uqt = ymin + i * dy;
    c += h[i];
  }
}
inline double dy_of(const int*) { return 1; }
inline double dy_of(const float*) { return 0.01; }
inline double dy_of(const double*) { return 0.01; }
template <class T> double pmedian(const T* x, size_t n) { double a, b, c; pstat(x, n, dy_of(x), a, b, c); return b; }
template <class T> double piqr(const T* x, size_t n) { double a, b, c; pstat(x, n, dy_of(x), a, b, c); return c - a; }

// variancetp with end_rule -1, wufunctions.cpp:766-809: E[x^2]-E[x]^2 with double accumulators.
template <class T>
double variance(const T* y, int n) {
  double sum = 0.0, sum2 = 0.0;
  for (int i = 0; i < n; ++i) { sum += (double)y[i]; sum2 += (double)y[i] * (double)y[i]; }
  double mean = sum / double(n);
  return sum2 / double(n) - mean * mean;
}
// mean_tp, wufunctions.cpp:666-690
double mean_range(const std::vector<int>& y, int low, int high) {
  double sum = 0;
  for (int i = low; i <= high; ++i) sum += (double)y[i];
  return sum / double(high - low + 1);
}

// Exact sample median, alglib samplemedian (alglib/statistics.cpp:3237-3385) through
// alglibinterface.cpp:18-30: middle order statistic, or 0.5*(two middles) for even n.
double exact_median(const int* x, int n) {
  if (n <= 0) return 0;
  std::vector<double> v(x, x + n);
  int k = (n - 1) / 2;
  std::nth_element(v.begin(), v.begin() + k, v.end());
  if (n % 2 == 1) return v[k];
  double a = *std::min_element(v.begin() + k + 1, v.end());
  return 0.5 * (v[k] + a);
}

// Normal CDF, alglib normaldistribution/errorfunction/errorfunctionc
// (alglib/specialfunctions.cpp:3152-3302; Cephes ndtr rational approximations).
double erfc_cephes(double x);
double erf_cephes(double x) {
  static const double P[] = {0.007547728033418631287834, -0.288805137207594084924010, 14.3383842191748205576712,
                             38.0140318123903008244444, 3017.82788536507577809226, 7404.07142710151470082064,
                             80437.3630960840172832162};
  static const double Q[] = {0.0, 1.00000000000000000000000, 38.0190713951939403753468, 658.070155459240506326937,
                             6379.60017324428279487120, 34216.5257924628539769006, 80437.3630960840172826266};
  double s = x > 0 ? 1 : (x < 0 ? -1 : 0);
  x = fabs(x);
  if (x < 0.5) {
    double xsq = x * x, p = P[0], q = Q[0];
    for (int i = 1; i < 7; ++i) { p = P[i] + xsq * p; q = Q[i] + xsq * q; }
    return s * 1.1283791670955125738961589031 * x * p / q;
  }
  if (x >= 10) return s;
  return s * (1 - erfc_cephes(x));
}
double erfc_cephes(double x) {
  static const double P[] = {0.0, 0.5641877825507397413087057563, 9.675807882987265400604202961,
                             77.08161730368428609781633646, 368.5196154710010637133875746,
                             1143.262070703886173606073338, 2320.439590251635247384768711,
                             2898.0293292167655611275846, 1826.3348842295112592168999};
  static const double Q[] = {1.0, 17.14980943627607849376131193, 137.1255960500622202878443578,
                             661.7361207107653469211984771, 2094.384367789539593790281779,
                             4429.612803883682726711528526, 6089.5424232724435504633068,
                             4958.82756472114071495438422, 1826.3348842295112595576438};
  if (x < 0) return 2 - erfc_cephes(-x);
  if (x < 0.5) return 1.0 - erf_cephes(x);
  if (x >= 10) return 0;
  double p = P[0], q = Q[0];
  for (int i = 1; i < 9; ++i) { p = P[i] + x * p; q = Q[i] + x * q; }
  return exp(-(x * x)) * p / q;
}
double pnorm(double x) { return 0.5 * (erf_cephes(x / 1.41421356237309504880) + 1); }

// ---------------------------------------------------------------------------------------------
struct Oracle {
  mutable int short_neighbourhoods = 0;   // tests the reference would have aborted in (isitcnv)
  Params P;
  int n = 0;                       // chromosome length
  std::vector<Region> noncode;     // padded N regions
  std::vector<int> rd;             // current per-base depth (shrinks at compaction)
  std::vector<uint8_t> gc;
  double gc_table[202];
  double gc_rdmean = 0;
  double cap_median = 0;           // median used by apply_cap
  double RDmedian = 0, RDsd = 0;
  int start = 1, end = 0;          // rsi::start / rsi::end
  // bins
  std::vector<float> binmed, binnb;
  std::vector<int> binmedint;
  double nb_median = 0, nb_mad = 0, nb_r = 0, nb_tmin = 0;
  double factor = 0;
  int LmaxBase = 0;
  // scan
  struct Scan {
    double tmedian1 = 0, tsigma1 = 0, tlamda1 = 0, tmedian2 = 0, tsigma2 = 0, tlamda2 = 0, target = 0;
    int Lmax = 0, cal_max = 0;
    std::vector<int> st1, st1f, st2;
    std::vector<Cnv> segs;
    int trim_escapes = 0;
  } scan_nb, scan_med;
  std::vector<Cnv> blocks;         // after areblockscnv (+ sort) in bin space
  std::vector<Cnv> calls_raw, calls;
  std::map<std::string, std::vector<int>> snap_i;   // named snapshots of per-base arrays
  double stage_s[8] = {0};

  // ---- A1: GC mask + padded N regions: loaddata.cpp:481-486, 243-273; readref.cpp:88-112 ----
  void load(const Params& p, const int32_t* depth, const uint8_t* fasta, int n_) {
    P = p; n = n_;
    gc.resize(n);
    for (int k = 0; k < n; ++k) gc[k] = (fasta[k] == 'G' || fasta[k] == 'C');
    // maximal runs of upper-case 'N', padded by dx each side, clamped, then re-merged when they
    // overlap or touch (the reference paints the padding as 'N' and scans again)
    int dx = std::max(50, P.m / 4);
    std::vector<Region> runs;
    for (int i = 0; i < n;) {
      if (fasta[i] != 'N') { ++i; continue; }
      int j = i;
      while (j + 1 < n && fasta[j + 1] == 'N') ++j;
      runs.push_back({std::max(0, i - dx), std::min(n - 1, j + dx)});
      i = j + 1;
    }
    noncode.clear();
    for (const Region& r : runs) {
      if (!noncode.empty() && r.start <= noncode.back().end + 1) noncode.back().end = std::max(noncode.back().end, r.end);
      else noncode.push_back(r);
    }
    rd.assign(depth, depth + n);
    start = 1; end = n;
    snap_i.clear();
  }

  // window GC count of pass 1 for base i (gccontent.cpp:124-133)
  int window_q1(const std::vector<int>& pg, int i) const {
    int lo, hi;
    if (i <= 100) { lo = 0; hi = 200; }
    else if (i <= n - 102) { lo = i - 100; hi = i + 100; }
    else { lo = n - 202; hi = n - 2; }
    return pg[hi + 1] - pg[lo];
  }
  static int rescale(int v, double rdmean, double denom) { return (int)(v * rdmean / denom + 0.5); }   // gccontent.cpp:89

  // ---- A2 + A3: checkgccontent / adjustgccontent, gccontent.cpp:95-184, 43-92 ----
  void gc_correct() {
    if (!P.gcadjust) return;
    double sum = 0.0; int cnt = 0;
    for (int i = 0; i < n; ++i) if (rd[i] > 0) { sum += rd[i]; ++cnt; }
    double rdmean = cnt > 0 ? sum / (double)cnt : sum;
    std::vector<int> pg(n + 1, 0);
    for (int i = 0; i < n; ++i) pg[i + 1] = pg[i] + gc[i];
    double tsum[202]; int tcnt[202];
    for (int g = 0; g < 202; ++g) { tsum[g] = 0; tcnt[g] = 0; }
    for (int i = 0; i < n; ++i) { int g = window_q1(pg, i); tsum[g] += rd[i]; tcnt[g]++; }
    for (int g = 0; g < 202; ++g) {
      gc_table[g] = tcnt[g] > 0 ? tsum[g] / double(tcnt[g]) : rdmean;
      if (gc_table[g] < 1) gc_table[g] = rdmean;
    }
    gc_rdmean = rdmean;
    // pass 2: 20 slices of S = n/20; every slice start recomputes the true clamped window, which
    // equals the pass-1 window everywhere below 20*S; the tail follows App. A Q2/Q3.
    const int S = n / 20, r = n - 20 * S;
    std::vector<int> out(rd);
    for (int i = 0; i < 20 * S; ++i) out[i] = rescale(rd[i], rdmean, gc_table[window_q1(pg, i)]);
    if (r >= 2) {
      int gtail = pg[n] - pg[n - 201];   // fresh edge window [n-201, n-1]
      for (int k = 0; k < r; ++k) out[n - 201 + k] = rescale(rd[20 * S + k], rdmean, gc_table[gtail]);
    }
    rd.swap(out);   // the last r bases keep their unadjusted depth
  }

  // ---- A4: apply_cap, loaddata.cpp:229-240 ----
  void cap() {
    if (P.cap <= 1) return;
    cap_median = pmedian(rd.data(), rd.size());
    double c = cap_median * P.cap;
    for (int i = 0; i < n; ++i) if (rd[i] > c) rd[i] = (int)c;
  }

  // ---- A5 + A6: concatenate_data (loaddata.cpp:48-85) and rsi.cpp:2202-2203 ----
  void concat() {
    if (!noncode.empty()) {
      std::vector<char> drop(rd.size(), 0);
      for (const Region& r : noncode) for (int i = r.start; i <= r.end; ++i) drop[i] = 1;
      size_t w = 0;
      for (size_t i = 0; i < rd.size(); ++i) if (!drop[i]) rd[w++] = rd[i];
      rd.resize(w);
      start = 1; end = (int)rd.size();
    }
    RDmedian = pmedian(rd.data(), rd.size());
    RDsd = sqrt(variance(rd.data(), (int)rd.size()));
  }

  // ---- A8: median_transfer + RDmedint, rsi.cpp:1363-1379, 1816-1819 ----
  // ---- A9 + A10: negative_binomial_transfer, rsi.cpp:1120-1188 ----
  static double nb_formula(double sum, double m2, double r) {
    return 2.0 * sqrt(r) * log(sqrt((sum + 0.25) / (m2 * r - 0.5)) + sqrt(1.0 + (sum + 0.25) / (m2 * r - 0.5)));
  }
  void bins() {
    const int m = P.m, np = (int)rd.size(), nb = np / m;
    binmed.assign(nb, -1.0f); binmedint.assign(nb, 0); binnb.assign(nb, -1.0f);
    for (int b = 0; b < nb; ++b) {
      binmed[b] = (float)exact_median(&rd[(size_t)b * m], m);
      binmedint[b] = (int)(binmed[b] + 0.5);
    }
    RDmedian = pmedian(rd.data(), rd.size());   // rsi.cpp:1821
    // MAD over 31 interleaved subsamples
    double med = pmedian(rd.data(), rd.size());
    const int ns = 31, len = np / ns;
    std::vector<int> tmp(len);
    double mads[31];
    for (int j = 0; j < ns; ++j) {
      for (int k = 0, i = j; k < len && i < np; ++k, i += ns) tmp[k] = (int)fabs((float)rd[i] - med);
      mads[j] = pmedian(tmp.data(), tmp.size());
    }
    double mad = pmedian(mads, (size_t)ns);
    double r = med / mad;
    nb_median = med; nb_mad = mad; nb_r = r;
    for (int b = 0; b < nb; ++b) {
      int i1 = b * m, i2 = std::min(b * m + m - 1, np - 1);
      double m2 = double(i2 - i1 + 1), sum = 0.0;
      for (int j = i1; j <= i2; ++j) sum += rd[j];
      binnb[b] = (float)nb_formula(sum, m2, r);
    }
    double med_nbt = nb_formula(med * m, (double)m, r);
    double del_nbt = nb_formula(med / 2.0 * (double)m, (double)m, r);
    double dup_nbt = nb_formula(med * 1.5 * (double)m, (double)m, r);
    double tmin = binnb[0];
    for (int b = 0; b < nb; ++b) if (binnb[b] < tmin) tmin = binnb[b];
    for (int b = 0; b < nb; ++b) binnb[b] = (float)(binnb[b] - tmin);
    med_nbt -= tmin;
    for (int b = 0; b < nb; ++b) binnb[b] = (float)(binnb[b] / med_nbt * med);
    del_nbt -= tmin; dup_nbt -= tmin;
    del_nbt = del_nbt / med_nbt * med;
    dup_nbt = dup_nbt / med_nbt * med;
    med_nbt = med_nbt / med_nbt * med;
    binnb[0] = (float)del_nbt; binnb[1] = (float)dup_nbt; binnb[2] = (float)med_nbt;   // App. A Q9
    nb_tmin = tmin;
    factor = sqrt(2.0 * (1.0 + P.epsilon) * log(3.1E9));   // rsi.cpp:1829
    LmaxBase = std::max(20, 10000 / m);                    // rsi.cpp:1830-1831
  }

  // ---- A12: rsistatus (rsi.cpp:1191-1259) with runmeantp (wufunctions.cpp:573-647) ----
  void runmean(const std::vector<float>& y, std::vector<float>& smo, int L) const {
    const int nb = (int)y.size();
    double sum = 0;
    for (int i = 0; i < L; ++i) sum += (double)y[i];
    double mean = sum / double(L);
    int h = L / 2;
    for (int i = 0; i <= h; ++i) smo[i] = (float)mean;
    int ismo = h + 1;
    for (int first = 1, last = L; last < nb; ++first, ++last, ++ismo) {
      sum = sum - (double)y[first - 1] + (double)y[last];
      mean = sum / double(L);
      smo[ismo] = (float)mean;
    }
    for (int i = ismo; i < nb; ++i) smo[i] = (float)mean;
  }
  void rsistatus(const std::vector<float>& T, double tmedian, double tlamda, int Lmax, std::vector<int>& st, int& escapes) const {
    const int nb = (int)T.size();
    st.assign(nb, 0);
    std::vector<float> smo(nb, 0.0f);
    for (int sweep = 0; sweep < 2; ++sweep) {       // 0: deletions, 1: duplications
      const bool del = sweep == 0;
      const double lim = del ? RDmedian * 0.75 : RDmedian * 1.25;
      for (int L = 1; L <= Lmax; ++L) {
        if (L > nb) break;   // the reference exits the process here (wufunctions.cpp:591-598)
        runmean(T, smo, L);
        for (int i = L / 2 + 1; i < nb - L / 2 - 1; ++i) {
          double score = (smo[i] - tmedian) * sqrt(double(L));
          if (del ? (score > -tlamda) : (score < tlamda)) continue;
          int i1 = i - L / 2, i2 = i1 + L - 1;
          double wm = exact_median(&binmedint[i1], L);
          if (del ? (wm > lim) : (wm < lim)) continue;
          // trims, in the reference's order, each an unbounded walk (App. A Q12)
          bool ok = true;
          auto off = [&](int q) { return q < 0 || q >= nb; };
          if (del) {
            while (ok && T[i1] > tmedian) { if (off(++i1)) ok = false; }
            while (ok && binmedint[i1] > lim) { if (off(++i1)) ok = false; }
            while (ok && T[i2] > tmedian) { if (off(--i2)) ok = false; }
            while (ok && binmedint[i2] > lim) { if (off(--i2)) ok = false; }
          } else {
            while (ok && T[i1] < tmedian) { if (off(++i1)) ok = false; }
            while (ok && binmedint[i1] < lim) { if (off(++i1)) ok = false; }
            while (ok && T[i2] < tmedian) { if (off(--i2)) ok = false; }
            while (ok && binmedint[i2] < lim) { if (off(--i2)) ok = false; }
          }
          if (!ok) { ++escapes; continue; }   // the reference would abort on the bounds check
          for (int j = i1; j <= i2; ++j) if (st[j] == 0) st[j] = del ? -L : L;
        }
        int icount = 0;
        for (int i = 0; i < nb; ++i) icount += del ? (st[i] < 0) : (st[i] > 0);
        if (double(icount) / double(nb) > 0.2) break;
      }
    }
  }

  // ---- get_continuous_segments, rsi.cpp:291-327 (the final run is never emitted, Q11) ----
  template <class S>
  static void runs_of(const S& st, int d, std::vector<Cnv>& out) {
    out.clear();
    int istart = 0, iend = 0, icount = 0;
    for (int i = 0; i < (int)st.size(); ++i) {
      if (st[i] == 0) continue;
      if (icount == 0) { istart = iend = i; icount++; continue; }
      if ((double)st[i] * (double)st[iend] > 0 && (i - iend) <= d) { iend = i; continue; }
      Cnv s; s.start = istart; s.end = iend;
      out.push_back(s);
      istart = iend = i; icount++;
    }
  }

  // ---- A13: filterstatus_tp, rsi.cpp:948-1047 ----
  void filterstatus(const std::vector<float>& T, double dev, std::vector<int>& st) const {
    const int nb = (int)T.size();
    int minlevel = st[0], maxlevel = st[0];
    for (int i = 0; i < nb; ++i) { minlevel = std::min(minlevel, st[i]); maxlevel = std::max(maxlevel, st[i]); }
    int nl = maxlevel - minlevel + 1;
    std::vector<float> lsum(nl, 0.0f);
    std::vector<int> lcnt(nl, 0);
    for (int i = 0; i < nb; ++i) { lsum[st[i] - minlevel] += T[i]; ++lcnt[st[i] - minlevel]; }   // float accumulators
    for (int l = 0; l < nl; ++l) if (lcnt[l] != 0) lsum[l] /= (double)lcnt[l];
    if (-minlevel < 0 || -minlevel >= nl) return;   // no unmarked bin: the reference would throw
    float m0 = lsum[-minlevel];
    int leveldel = minlevel, leveladd = maxlevel;
    for (int l = 0; l < nl; ++l) if (lsum[l] < m0 - dev) { leveldel = l + minlevel; break; }
    for (int l = nl - 1; l >= 0; --l) if (lsum[l] > m0 + dev) { leveladd = l + minlevel; break; }
    if (leveldel > 0 || leveladd < 0 || leveldel > leveladd) return;
    double delthr = m0 - dev, addthr = m0 + dev;
    std::vector<Cnv> segs;
    runs_of(st, 1, segs);
    for (const Cnv& s : segs) {
      int i1 = s.start, i2 = s.end;
      while ((T[i1] > delthr && st[i1] < 0) || (T[i1] < addthr && st[i1] > 0)) { st[i1] = 0; ++i1; if (i1 >= i2) break; }
      while ((T[i2] > delthr && st[i2] < 0) || (T[i2] < addthr && st[i2] > 0)) { st[i2] = 0; --i2; if (i2 <= i1) break; }
    }
  }

  // ---- A14: get_rsi_segments, rsi.cpp:1060-1117 ----
  void rsi_segments(const std::vector<float>& T, const std::vector<int>& st, double tmedian, std::vector<Cnv>& out) const {
    std::vector<Cnv> runs;
    runs_of(st, 1, runs);
    for (const Cnv& run : runs) {
      int a = run.start, b = run.end, bs = a, be = b;
      double best = 0;
      for (int L = 1; L <= b - a + 1; ++L) {
        double sum = 0.0;
        for (int j = a; j <= b && j < a + L; ++j) sum += T[j];
        for (int j = a;; ++j) {
          double score = fabs(sum / (double)L - tmedian) * sqrt(double(L));
          if (score > best) { bs = j; be = j + L - 1; best = score; }
          if (j + L > b) break;
          sum = sum - T[j] + T[j + L];
        }
      }
      Cnv c; c.start = bs; c.end = be;
      if (pmedian(&st[bs], (size_t)(be - bs + 1)) > 0) { c.type = T_DUP; c.score = best; }
      else { c.type = T_DEL; c.score = -best; }
      out.push_back(c);
    }
  }

  // ---- A11 + driver: rsicnvnbn (rsi.cpp:1262-1360) / rsicnvmed (rsi.cpp:1402-1501) ----
  void scan(bool use_med, Scan& R) {
    const std::vector<float>& T = use_med ? binmed : binnb;
    const int nb = (int)T.size();
    std::vector<float> tmp(nb);
    double tmedian, tsigma, tlamda, target, dev;
    int Lmax = LmaxBase, cal_max;
    if (!use_med) {
      tmedian = pmedian(T.data(), T.size());
      for (int i = 0; i < nb; ++i) tmp[i] = (float)fabs(T[i] - tmedian);
      tsigma = pmedian(tmp.data(), tmp.size()) / 0.6745;
      tlamda = factor * tsigma;
      target = (T[2] - T[0]) * sqrt(2.5);
      tlamda = std::max(tlamda, target);
      double dnb = fabs(T[2] - T[0]) + 0.0001;
      double q = tlamda * 2 / dnb;
      cal_max = (int)(q * q);
      dev = tsigma * 3.0;
    } else {
      tmedian = RDmedian;
      for (int i = 0; i < nb; ++i) tmp[i] = (float)fabs(T[i] - tmedian);
      tsigma = pmedian(tmp.data(), tmp.size()) / 0.6745;
      tlamda = factor * tsigma;
      target = tmedian * sqrt(2.0);
      tlamda = std::max(tlamda, target);
      if (P.threshold > 0) tlamda = tmedian * P.threshold;
      double q = tlamda * 4 / (tmedian + 0.001);
      cal_max = (int)(q * q);
      dev = tmedian * 0.6;
    }
    if (Lmax < cal_max) Lmax = cal_max;
    R.tmedian1 = tmedian; R.tsigma1 = tsigma; R.tlamda1 = tlamda; R.target = target; R.Lmax = Lmax; R.cal_max = cal_max;
    R.trim_escapes = 0;
    rsistatus(T, tmedian, tlamda, Lmax, R.st1, R.trim_escapes);
    R.st1f = R.st1;
    filterstatus(T, dev, R.st1f);
    int k = 0;
    for (int i = 0; i < nb; ++i) if (R.st1f[i] == 0) tmp[k++] = T[i];
    if (k > nb / 2) {
      tmedian = pmedian(tmp.data(), (size_t)k);
      for (int i = 0; i < k; ++i) tmp[i] = (float)fabs(tmp[i] - tmedian);
      tsigma = pmedian(tmp.data(), (size_t)k) / 0.6745;
      tlamda = factor * tsigma;
      tlamda = std::max(tlamda, target);
    }
    R.tmedian2 = tmedian; R.tsigma2 = tsigma; R.tlamda2 = tlamda;
    rsistatus(T, tmedian, tlamda, Lmax, R.st2, R.trim_escapes);
    std::vector<Cnv> segs;
    rsi_segments(T, R.st2, tmedian, segs);
    R.segs.clear();
    for (Cnv& s : segs) if (!(fabs(s.score) < tlamda * 0.5)) R.segs.push_back(s);
  }

  // ---- isitcnv, rsi.cpp:101-172 ----
  void isitcnv(const std::vector<int>& ref, const std::vector<int>& cnv, Cnv& c) const {
    int d = (int)cnv.size(), nr = (int)ref.size() - d;
    if (nr <= 0) {   // a neighbourhood no longer than the candidate: the reference sizes its running-mean array RDref.size() - d
                     // (rsi.cpp:107) and indexes [0]: its Array throws, the program aborts.  Counted; orc_run reports it (-3).
      ++short_neighbourhoods; c.geno = 0; c.status = -9; return;
    }
    std::vector<float> rm(nr > 0 ? nr : 0);
    double sum = 0;
    for (int i = 0; i < d; ++i) sum += ref[i];
    if (nr > 0) rm[0] = (float)(sum / double(d));
    for (int i = 1; i < nr; ++i) { sum = sum - ref[i - 1] + ref[i - 1 + d]; rm[i] = (float)(sum / double(d)); }
    double rmed = pmedian(rm.data(), rm.size());
    double rsd = sqrt(variance(rm.data(), (int)rm.size()));
    if (rsd < 1E-3) rsd = rmed / 40.0 + 1E-3;
    c.length = c.end - c.start + 1;
    c.cnvmed = pmedian(cnv.data(), cnv.size());
    c.cnvsd = sqrt(variance(cnv.data(), (int)cnv.size()));
    c.cnviqr = piqr(cnv.data(), cnv.size());
    c.refmed = rmed;
    c.refsd = piqr(rm.data(), rm.size()) / 1.349;
    c.refiqr = piqr(rm.data(), rm.size());
    c.geno = 1; c.status = 1;
    int flag = c.cnvmed > RDmedian ? T_DUP : T_DEL;
    if (c.type == T_UNKNOWN) c.type = flag;
    if (c.type != flag) { c.status = -9; return; }
    if (c.type == T_DEL) {
      double reference = std::min(rmed, RDmedian);
      reference = std::max(reference, 0.8 * RDmedian);
      double nu = (3.0 * c.cnvmed - 2.0 * reference) / rsd;
      c.p1 = pnorm(nu);
      if (nu > 0) { c.status = -9; c.geno = 0; }
    } else {
      double reference = std::max(rmed, RDmedian);
      double nu = (2.5 * c.cnvmed - 3.0 * reference) / rsd / 1.5;
      c.p1 = 1.0 - pnorm(nu);
      if (nu < 0) { c.status = -9; c.geno = 0; }
    }
  }

  // ---- isitcnvwrap, rsi.cpp:175-287: gather the neighbourhood, then isitcnv ----
  void isitcnvwrap(const std::vector<int>& RD, std::vector<Cnv>& L, int ci) const {
    const int N = (int)RD.size(), nL = (int)L.size();
    const int flag = L[ci].type;
    const int cnvlen = L[ci].end - L[ci].start + 1;
    const int pts = P.maxchkbp * 10;
    int d = cnvlen;
    if (N == end - start + 1) { if (d < P.m * P.minmlen) d = (int)(P.m * P.minmlen); }
    if (N < (end - start + 1) / 2) { if (d < P.minmlen) d = (int)P.minmlen + 1; }
    std::vector<int> ref((size_t)(int)(P.chklen * d * 2), 0);
    const int refsize = (int)ref.size();
    const double upper = 3.0, lower = 0.15;
    auto extreme = [&](int v) { return (flag == T_DEL && v > RDmedian * upper) || (flag == T_DUP && v < RDmedian * lower); };

    int buffer = int(cnvlen * P.buffer + 1);
    int i = L[ci].start - buffer, idx = ci - 1;
    while (i > 0 && idx > 0 && i < L[idx].start) --idx;
    while (idx > 0 && L[idx].status == -9) --idx;
    int k = (int)(P.chklen * d - 1);
    if (N - L[ci].end < P.chklen * d) k = refsize - 1 - N + L[ci].end;
    const int stopper = k;
    while (i > 2 && k >= 0) {
      --i;
      if (extreme(RD[i])) continue;
      if (idx >= 0 && i >= L[idx].start && i <= L[idx].end) {
        i = L[idx].start - 1; --idx;
        while (idx > 0 && L[idx].status == -9) --idx;
        continue;
      }
      ref[k] = RD[i]; --k;
    }
    if (k >= 0) { int w = 0; for (int q = k + 1; q <= stopper; ++q) ref[w++] = ref[q]; k = w; }
    else k = stopper + 1;

    i = L[ci].end + buffer; idx = ci + 1;
    while (i < N - 2 && idx < nL && i > L[idx].end) ++idx;
    while (idx < nL - 1 && L[idx].status == -9) ++idx;
    while (i < N - 2 && k < 2 * P.chklen * d) {
      ++i;
      if (extreme(RD[i])) continue;
      if (idx < nL && i >= L[idx].start && i <= L[idx].end) {
        i = L[idx].end + 1; ++idx;
        while (idx < nL - 1 && L[idx].status == -9) ++idx;
        continue;
      }
      if (k >= refsize) break;   // cannot happen with the default -reflen (5*d is an integer)
      ref[k] = RD[i]; ++k;
    }
    if (k < refsize) ref.resize(k);
    std::vector<int> cnv(RD.begin() + L[ci].start, RD.begin() + L[ci].end + 1);
    int tot = (int)ref.size() + (int)cnv.size();
    if (tot > pts) {   // thin both to about pts points in all (rsi.cpp:264-282)
      int dref = (int)((double)ref.size() / (double)tot * (double)pts);
      int dcnv = (int)((double)cnv.size() / (double)tot * (double)pts);
      std::vector<int> t(dref);
      for (int q = 0; q < dref; ++q) t[q] = ref[(int)(double(q) / double(dref) * double(ref.size()))];
      ref.swap(t);
      t.assign(dcnv, 0);
      for (int q = 0; q < dcnv; ++q) t[q] = cnv[(int)(double(q) / double(dcnv) * double(cnv.size()))];
      cnv.swap(t);
    }
    isitcnv(ref, cnv, L[ci]);
  }

  // ---- multisegments, rsi.cpp:368-410 ----
  static void multisegments(const Cnv& seg, const std::vector<int>& st, std::vector<Cnv>& out) {
    out.clear();
    std::vector<int> s2(st.begin() + seg.start, st.begin() + seg.end + 1), flag(s2.size(), 0);
    int lo = *std::min_element(s2.begin(), s2.end()), hi = *std::max_element(s2.begin(), s2.end());
    for (int level = lo; level < hi; ++level) {
      if (level == 0) continue;
      std::fill(flag.begin(), flag.end(), 0);
      int levelcount = 0;
      for (size_t i = 0; i < s2.size(); ++i) {
        if (s2[i] == 0) continue;
        if (s2[i] == level) ++levelcount;
        if (level < 0 && s2[i] < 0 && s2[i] >= level) flag[i] = 1;
        if (level > 0 && s2[i] > 0 && s2[i] <= level) flag[i] = 1;
      }
      if (levelcount == 0) continue;
      std::vector<Cnv> sub;
      runs_of(flag, 1, sub);
      for (Cnv& s : sub) {
        s.start += seg.start; s.end += seg.start;
        if (s.start < seg.start) s.start = seg.start;
        if (s.end > seg.end) s.end = seg.end;
        out.push_back(s);
      }
    }
  }

  // ---- A15: areblockscnv, rsi.cpp:415-546 ----
  void areblockscnv(const std::vector<int>& st, std::vector<Cnv>& segs) const {
    std::vector<Cnv> T = segs;
    for (int i = 0; i < (int)T.size(); ++i) isitcnvwrap(binmedint, T, i);
    for (int i = 0; i < (int)T.size(); ++i) {
      if (T[i].status != -9) continue;
      if (T[i].type == T_DEL && T[i].cnvmed < 0.7 * T[i].refmed) { T[i].geno = 1; T[i].p1 = P.p; continue; }
      if (T[i].type == T_DUP && T[i].cnvmed > 1.3 * T[i].refmed) { T[i].geno = 1; T[i].p1 = P.p; continue; }
      Cnv oseg = T[i], iseg = T[i];
      std::vector<Cnv> ms;
      multisegments(iseg, st, ms);
      for (int j = (int)ms.size() - 1; j >= 0; --j) {
        ms[j].type = iseg.type;
        T[i] = ms[j];
        isitcnvwrap(binmedint, T, i);
        ms[j] = T[i];
      }
      for (int j = (int)ms.size() - 1; j >= 0; --j) {
        if (ms[j].geno == 0) continue;
        if (iseg.geno == 0) iseg = ms[j];
        if (ms[j].length > iseg.length) iseg = ms[j];
      }
      if (iseg.geno == 0) iseg = oseg;
      T[i] = iseg;
    }
    segs = T;
  }

  // ---- A17: sortcnvstartposition, rsi.cpp:549-577 (stable; arrayindex_tp wufunctions.cpp:1520) ----
  static void sort_by_start(std::vector<Cnv>& L) {
    for (Cnv& c : L) if (c.start > c.end) std::swap(c.start, c.end);
    std::stable_sort(L.begin(), L.end(), [](const Cnv& a, const Cnv& b) { return a.start < b.start; });
  }

  // ---- A16: optimize_with_derivative, rsi.cpp:889-944 ----
  void refine_edges(const std::vector<int>& RD, Cnv& c) const {
    int len = c.end - c.start + 1, disp = std::max(250, len / 4);
    int nstart = c.start - disp, nend = c.end + disp;
    if (nstart < 2 * len) return;
    if (nend > (int)RD.size() - 2 * len) return;
    std::vector<double> dd;
    double diff = 0.0;
    for (int k = nstart - len; k < nstart; ++k) diff += RD[k];
    for (int k = nstart; k < nstart + len; ++k) diff -= RD[k];
    dd.push_back(diff);
    for (int i = nstart + 1; i < nend; ++i) {
      diff = diff - RD[i - 1 - len] + RD[i - 1] + RD[i - 1] - RD[i - 1 + len];
      dd.push_back(diff);
    }
    // A candidate whose first refinement left end < start (len <= 0) has fewer than 2 * disp entries: the reference's loops
    // (rsi.cpp:917, 930) then index its vector out of range -- undefined behaviour, heap contents decide.  The restatement
    // (and the library, host_calls.cpp:sharpen_edges / k_sharpen_edges) searches the entries that exist; parity on such a
    // candidate cannot be defined (divergences).
    const int nd = (int)dd.size();
    int imax = -1; double best = 0;
    for (int i = 0; i < 2 * disp && i < nd; ++i) {
      if (c.type == T_DEL && dd[i] > best) { best = dd[i]; imax = i; }
      if (c.type == T_DUP && dd[i] < best) { best = dd[i]; imax = i; }
    }
    if (imax > 0) c.start = nstart + imax;
    imax = -1; best = 0;
    for (int i = std::max(0, nd - 2 * disp); i < nd; ++i) {
      if (c.type == T_DEL && dd[i] < best) { best = dd[i]; imax = i; }
      if (c.type == T_DUP && dd[i] > best) { best = dd[i]; imax = i; }
    }
    if (imax > 0) c.end = nend - (int)dd.size() + imax;
  }

  // ---- A17: mergesegments, rsi.cpp:694-885 ----
  void mergesegments(const std::vector<int>& RD, std::vector<Cnv>& L) const {
    std::vector<Cnv> T;
    for (int i = 0; i < (int)L.size() - 1; ++i) {
      if (L[i].type != L[i + 1].type) continue;
      if (!(std::max(L[i].start, L[i + 1].start) < std::min(L[i].end, L[i + 1].end))) continue;
      Cnv u = L[i];
      u.start = std::min(L[i].start, L[i + 1].start);
      u.end = std::max(L[i].end, L[i + 1].end);
      T = L; T[i] = u; T[i + 1] = u; T[i + 1].status = -9;
      isitcnvwrap(RD, T, i);
      if (T[i].geno == 0) {
        T = L; T[i + 1].status = -9;
        isitcnvwrap(RD, T, i);
        T[i].status = -9; T[i + 1].status = 0;
        isitcnvwrap(RD, T, i + 1);
        if (T[i + 1].p1 < T[i].p1) T[i] = T[i + 1];
        if (T[i].p1 > P.p) { L[i].status = -9; L[i + 1].status = -9; }
      }
      if (T[i].geno == 0) continue;
      L[i] = T[i]; L[i].status = -9;
      L[i + 1] = T[i]; L[i + 1].status = 0;
    }
    std::vector<Cnv> K;
    for (const Cnv& c : L) if (c.status != -9) K.push_back(c);
    L = K;
    if (!P.merge) return;
    for (int i = 0; i < (int)L.size() - 1; ++i) {
      if (L[i].type != L[i + 1].type) continue;
      if (L[i].geno == 0 || L[i + 1].geno == 0) continue;
      int gap = L[i + 1].start - L[i].end;
      if (gap > (L[i].end - L[i].start) * P.chklen * 0.7 && gap > (L[i + 1].end - L[i + 1].start) * P.chklen * 0.7) continue;
      double m1 = mean_range(RD, L[i].start, L[i].end), m2 = mean_range(RD, L[i + 1].start, L[i + 1].end);
      double cm = (m1 * (L[i].end - L[i].start) + m2 * (L[i + 1].end - L[i + 1].start)) /
                  ((L[i].end - L[i].start) + (L[i + 1].end - L[i + 1].start));
      double mm = mean_range(RD, L[i].start, L[i + 1].end);
      if (L[i].type == T_DEL && mm > cm + 1.5 * L[i + 1].refsd + 1.5 * L[i].refsd) continue;
      if (L[i].type == T_DUP && mm < cm - 1.5 * L[i + 1].refsd - 1.5 * L[i].refsd) continue;
      Cnv u = L[i]; u.end = L[i + 1].end;
      T = L; T[i] = u; T[i + 1] = u; T[i + 1].status = -9;
      isitcnvwrap(RD, T, i);
      if (T[i].geno == 0) continue;
      L[i] = T[i]; L[i + 1] = T[i]; L[i].status = -9;
    }
    K.clear();
    for (const Cnv& c : L) if (c.status != -9) K.push_back(c);
    L = K;
  }

  // ---- expand_coordinate, rsi.cpp:1524-1551 ----
  int expand(int p1) const {
    if (noncode.empty()) return p1;
    int dx = 0, prev_inc = 0;
    for (size_t i = 0; i < noncode.size(); ++i) {
      dx += noncode[i].end - noncode[i].start + 1;
      int brk = noncode[i].end + 1 - dx;
      if (p1 < brk) return p1 + prev_inc;
      prev_inc = dx;
    }
    return p1 + prev_inc;
  }

  // ---- detectcnv, rsi.cpp:1795-1945, then sd_filters, rsi.cpp:1753-1792 ----
  void detect() {
    calls_raw.clear(); calls.clear(); blocks.clear();
    if (RDmedian < 5) return;
    bins();
    std::vector<Cnv> segs;
    if (P.trans != 0) { scan(true, scan_med); segs = scan_med.segs; areblockscnv(scan_med.st2, segs); }
    if (P.trans == 0) { scan(false, scan_nb); segs = scan_nb.segs; areblockscnv(scan_nb.st2, segs); }
    if (P.trans == 2) {
      scan(false, scan_nb);
      std::vector<Cnv> s2 = scan_nb.segs;
      areblockscnv(scan_nb.st2, s2);
      segs.insert(segs.end(), s2.begin(), s2.end());
    }
    sort_by_start(segs);
    blocks = segs;
    const int m = P.m, np = (int)rd.size();
    std::vector<Cnv> L;
    for (Cnv c : segs) {
      if (c.geno == 0) continue;
      if (c.start == c.end) continue;
      c.start = c.start * m + m / 2;
      c.end = c.end * m + m / 2;
      if (c.start < 0) c.start = 0;
      if (c.end > np - 1) c.end = np - 1;
      c.length = c.end - c.start + 1;
      c.tid = 0;
      L.push_back(c);
    }
    for (Cnv& c : L) refine_edges(rd, c);
    for (Cnv& c : L) refine_edges(rd, c);
    sort_by_start(L);
    mergesegments(rd, L);
    sort_by_start(L);
    std::vector<Cnv> fin;
    for (int i = 0; i < (int)L.size(); ++i) {
      double len = double(L[i].end - L[i].start + 1) / double(m);
      isitcnvwrap(rd, L, i);
      L[i].score = (L[i].cnvmed - RDmedian) * sqrt(len);
      int p1 = expand(L[i].start), p2 = expand(L[i].end);
      for (const Region& r : noncode) if (std::max(p1, r.start) <= std::min(p2, r.end)) L[i].status = -9;
      if (L[i].status != -9) fin.push_back(L[i]);
    }
    for (Cnv& c : fin) { c.start = expand(c.start); c.end = expand(c.end); }
    calls_raw = fin;
    // sd_filters
    int minlen = std::max(m * 2, 500);
    double tsd = RDsd / 1.2;
    for (const Cnv& c : fin) {
      bool keep = true;
      int span = abs(c.end - c.start);
      if (span < 1000) keep = false;
      if (c.type == T_DEL) {
        if (c.p1 > 0.2) keep = false;
        if (c.refsd > 0.6 * tsd) keep = false;
        if (c.cnvsd > 1.3 * tsd) keep = false;
        if (c.cnvsd * RDmedian > 2.5 * c.cnvmed * tsd) keep = false;
        if (c.cnvmed < 0.66 * std::min(RDmedian, c.refmed) && c.cnvsd < tsd && span > 800) keep = true;
      }
      if (c.type == T_DUP) {
        if (c.p1 > 0.05) keep = false;
        if (c.refsd > 0.6 * tsd) keep = false;
        if (c.cnvsd * RDmedian > 2.0 * c.cnvmed * tsd) keep = false;
      }
      if (span < minlen) keep = false;
      if (keep) calls.push_back(c);
    }
  }
};

double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

}  // namespace orc

// ---------------------------------------------------------------------------------------------
// C ABI (ctypes).  Layouts match oracle/ref_driver.cpp so tests compare like with like.
extern "C" {

struct orc_params { int32_t m, gcadjust, trans, merge, maxchkbp, debug; double cap, epsilon, threshold, chklen, minmlen, buffer, p; };
struct orc_call { int32_t start, end, type, geno, status, length, qscore, pad; double score, p1, cnvmed, cnvsd, cnviqr, refmed, refsd, refiqr; };

void* orc_create(void) { return new orc::Oracle(); }
void orc_destroy(void* h) { delete (orc::Oracle*)h; }
void orc_default_params(orc_params* p) {
  p->m = 101; p->gcadjust = 1; p->trans = 0; p->merge = 1; p->maxchkbp = 100000; p->debug = 0;
  p->cap = 4.0; p->epsilon = 1.5; p->threshold = -1.0; p->chklen = 2.5; p->minmlen = 3.01; p->buffer = 0.05; p->p = 0.05;
}
static orc::Params conv(const orc_params* p) {
  orc::Params q;
  q.m = p->m; q.gcadjust = p->gcadjust; q.trans = p->trans; q.merge = p->merge; q.maxchkbp = p->maxchkbp; q.debug = p->debug;
  q.cap = p->cap; q.epsilon = p->epsilon; q.threshold = p->threshold; q.chklen = p->chklen; q.minmlen = p->minmlen;
  q.buffer = p->buffer; q.p = p->p;
  return q;
}
// Runs the whole path, keeping snapshots of the per-base array after each stage.
// Returns the number of final calls; stage seconds are available through orc_get_f64("stage_s").
int orc_run(void* h, const orc_params* p, const int32_t* depth, const uint8_t* fasta, int32_t n, int32_t keep_snapshots) {
  orc::Oracle& O = *(orc::Oracle*)h;
  double t0 = orc::now_s();
  O.short_neighbourhoods = 0;
  O.load(conv(p), depth, fasta, n);
  double t1 = orc::now_s();
  O.gc_correct();
  if (keep_snapshots) O.snap_i["rd_gc"] = O.rd;
  double t2 = orc::now_s();
  O.cap();
  if (keep_snapshots) O.snap_i["rd_cap"] = O.rd;
  double t3 = orc::now_s();
  O.concat();
  double t4 = orc::now_s();
  O.detect();
  double t5 = orc::now_s();
  O.stage_s[0] = t1 - t0; O.stage_s[1] = t2 - t1; O.stage_s[2] = t3 - t2; O.stage_s[3] = t4 - t3; O.stage_s[4] = t5 - t4;
  if (O.short_neighbourhoods) return -3;   // the reference aborts on this input (a candidate longer than its neighbourhood)
  return (int)O.calls.size();
}

static const std::vector<int>* ivec(orc::Oracle& O, const char* name) {
  std::string s(name);
  if (s == "rd_concat") return &O.rd;
  if (s == "binmedint") return &O.binmedint;
  if (s == "nb_status1") return &O.scan_nb.st1;
  if (s == "nb_status1f") return &O.scan_nb.st1f;
  if (s == "nb_status2") return &O.scan_nb.st2;
  if (s == "med_status1") return &O.scan_med.st1;
  if (s == "med_status1f") return &O.scan_med.st1f;
  if (s == "med_status2") return &O.scan_med.st2;
  auto it = O.snap_i.find(s);
  return it == O.snap_i.end() ? nullptr : &it->second;
}
int64_t orc_get_i32(void* h, const char* name, int32_t* out, int64_t cap) {
  orc::Oracle& O = *(orc::Oracle*)h;
  std::string s(name);
  if (s == "noncode") {
    int64_t k = 0;
    for (auto& r : O.noncode) { if (k + 2 <= cap) { out[k] = r.start; out[k + 1] = r.end; } k += 2; }
    return k;
  }
  const std::vector<int>* v = ivec(O, name);
  if (!v) return -1;
  for (int64_t i = 0; i < (int64_t)v->size() && i < cap; ++i) out[i] = (*v)[i];
  return (int64_t)v->size();
}
int64_t orc_get_f32(void* h, const char* name, float* out, int64_t cap) {
  orc::Oracle& O = *(orc::Oracle*)h;
  std::string s(name);
  const std::vector<float>* v = s == "binmed" ? &O.binmed : (s == "binnb" ? &O.binnb : nullptr);
  if (!v) return -1;
  for (int64_t i = 0; i < (int64_t)v->size() && i < cap; ++i) out[i] = (*v)[i];
  return (int64_t)v->size();
}
int64_t orc_get_f64(void* h, const char* name, double* out, int64_t cap) {
  orc::Oracle& O = *(orc::Oracle*)h;
  std::string s(name);
  std::vector<double> v;
  if (s == "chrom") v = {O.RDmedian, O.RDsd, O.cap_median, O.gc_rdmean};
  else if (s == "gc_table") v.assign(O.gc_table, O.gc_table + 202);
  else if (s == "nb") v = {O.nb_median, O.nb_mad, O.nb_r, O.nb_tmin, O.factor, (double)O.LmaxBase};
  else if (s == "scan_nb" || s == "scan_med") {
    const orc::Oracle::Scan& R = s == "scan_nb" ? O.scan_nb : O.scan_med;
    v = {R.tmedian1, R.tsigma1, R.tlamda1, R.tmedian2, R.tsigma2, R.tlamda2, R.target, (double)R.Lmax, (double)R.cal_max,
         (double)R.trim_escapes};
  } else if (s == "stage_s") v.assign(O.stage_s, O.stage_s + 5);
  else return -1;
  for (int64_t i = 0; i < (int64_t)v.size() && i < cap; ++i) out[i] = v[i];
  return (int64_t)v.size();
}
static void fill(const orc::Cnv& c, orc_call* o) {
  o->start = c.start; o->end = c.end; o->type = c.type; o->geno = c.geno; o->status = c.status; o->length = c.length; o->pad = 0;
  double q1 = c.p1 < 1.0E-10 ? 99 : -10.0 * log(c.p1) / log(10.0);   // cnv_format1, rsi.cpp:583-585
  o->qscore = (int)q1;
  o->score = c.score; o->p1 = c.p1; o->cnvmed = c.cnvmed; o->cnvsd = c.cnvsd; o->cnviqr = c.cnviqr;
  o->refmed = c.refmed; o->refsd = c.refsd; o->refiqr = c.refiqr;
}
// which: "segs_nb", "segs_med", "blocks", "calls_raw", "calls"
int orc_get_calls(void* h, const char* which, orc_call* out, int32_t cap) {
  orc::Oracle& O = *(orc::Oracle*)h;
  std::string s(which);
  const std::vector<orc::Cnv>* L = s == "segs_nb" ? &O.scan_nb.segs : s == "segs_med" ? &O.scan_med.segs :
                                   s == "blocks" ? &O.blocks : s == "calls_raw" ? &O.calls_raw : s == "calls" ? &O.calls : nullptr;
  if (!L) return -1;
  for (size_t i = 0; i < L->size() && (int)i < cap; ++i) fill((*L)[i], &out[i]);
  return (int)L->size();
}

// numeric-utility probes (pinned against the reference in tests/test_oracle_vs_ref.py)
double orc_median_i32(const int32_t* x, int64_t n) { return orc::pmedian(x, (size_t)n); }
double orc_median_f32(const float* x, int64_t n) { return orc::pmedian(x, (size_t)n); }
double orc_median_f64(const double* x, int64_t n) { return orc::pmedian(x, (size_t)n); }
double orc_iqr_i32(const int32_t* x, int64_t n) { return orc::piqr(x, (size_t)n); }
double orc_iqr_f32(const float* x, int64_t n) { return orc::piqr(x, (size_t)n); }
double orc_exact_median_i32(const int32_t* x, int64_t n) { return orc::exact_median(x, (int)n); }
double orc_pnorm(double x) { return orc::pnorm(x); }

}  // extern "C"
