// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product path).
//
// C-ABI driver around the UNMODIFIED reference objects (compiled by oracle/Makefile.ref from
// /root/reference/src, where they lie).  It calls the reference's own functions stage by stage,
// in the order of the reference's per-chromosome loop (rsi.cpp:2189-2212) and of
// load_data_from_text after the parse loop (loaddata.cpp:478-531), and copies every intermediate
// out so that (a) golden fixtures can be generated (tools/make_golden.py) and (b) the CPU
// restatement in oracle/rsi_oracle.cpp and the HIP path can be compared against the real thing.
//
// Nothing here re-implements arithmetic: each stage is a call into the reference.  The one
// piece of orchestration that is mirrored (ref_scan_stepwise, following rsi.cpp:1262-1347 /
// 1402-1501) is self-checked against the reference's own rsicnvnbn/rsicnvmed on every call.
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>
#include <iostream>
#include <fstream>
#include <sstream>
#include <math.h>
#include <string>
#include <vector>
using namespace std;

#include "samfunctions.h"
#include "readref.h"
#include "alglibinterface.h"
#include "wu2.h"
#include "rsi.h"
#include "wufunctions.h"
#include "gccontent.h"
#include "loaddata.h"
#include "plotcnv.h"

// reference functions with external linkage but no header (rsi.cpp, loaddata.cpp)
void get_noseq_regions(string& FASTA);                                             // loaddata.cpp:243
void detectcnv(Array<int>& RD, vector<cnv_st>& cnvlist);                           // rsi.cpp:1795
void sd_filters(vector<cnv_st>& cnvlist);                                          // rsi.cpp:1753
void median_transfer(Array<int>& RD, int m, Array<float>& RDt);                    // rsi.cpp:1363
void negative_binomial_transfer(Array<int>& RD, int m, Array<float>& RDt);         // rsi.cpp:1120
void rsistatus(Array<float>& RDtrans, Array<int>& RDmedint, double tmedian, double tlamda,
               int Lmax, Array<int>& RDtrans_status);                              // rsi.cpp:1191
void filterstatus(Array<float>& RDtrans, double dev, Array<int>& RDtrans_status);  // rsi.cpp:1052
void get_rsi_segments(Array<float>& RDmed, Array<int>& RDstatus, double tmedian,
                      vector<cnv_st>& rsiseglist);                                 // rsi.cpp:1060
void rsicnvnbn(Array<float>& RDtrans, Array<int>& RDmedint, Array<int>& RDtrans_status,
               vector<cnv_st>& cnvlist);                                           // rsi.cpp:1262
void rsicnvmed(Array<float>& RDtrans, Array<int>& RDmedint, Array<int>& RDtrans_status,
               vector<cnv_st>& cnvlist);                                           // rsi.cpp:1402
void areblockscnv(Array<int>& RDmedint, Array<int>& RDtrans_status,
                  vector<cnv_st>& rsiseglist);                                     // rsi.cpp:415
string cnv_format1(cnv_st& icnv);                                                  // rsi.cpp:581
int expand_coordinate(int p1);                                                     // rsi.cpp:1524
void plot_icnv(cnv_st icnv, string title, string datfile, string gpfile, string imgfile);   // plotcnv.cpp:246

extern "C" {

struct ref_params {
  int32_t m;          // -m (odd)
  int32_t gcadjust;   // !-NOGC
  int32_t trans;      // 0 NBN (-NB), 1 MED (-MED), 2 ALL (-ALL)
  int32_t merge;      // !-nomerge
  int32_t maxchkbp;   // -maxchkbp
  int32_t debug;
  double cap;         // -cap
  double epsilon;     // -e
  double threshold;   // -threshold
  double chklen;      // -reflen
  double minmlen;
  double buffer;
  double p;
};

struct ref_call {
  int32_t start, end, type, geno, status, length, qscore, pad;
  double score, p1, cnvmed, cnvsd, cnviqr, refmed, refsd, refiqr;
};

struct ref_scan_scalars {
  double tmedian1, tsigma1, tlamda1;   // first pass
  double tmedian2, tsigma2, tlamda2;   // second pass
  double target_tlamda;
  int32_t Lmax, cal_max;
  int32_t stepwise_matches_reference;  // 1 when the mirrored sequence equals rsicnvnbn/rsicnvmed
  int32_t nseg;
};

}  // extern "C"

namespace {

struct Session {
  Array<int> RD;
  Array<bool> GC;
  Array<float> RDmed, RDnbn;
  Array<int> RDmedint;
  Array<int> st_pass1, st_filtered, st_final;
  vector<cnv_st> seglist;        // bin-space segments after the 0.5*tlamda filter
  vector<cnv_st> calls_raw;      // detectcnv output
  vector<cnv_st> calls;          // after sd_filters
  int saved_stderr;
  Session() : RD(1), saved_stderr(-1) {}
};
Session S;

void quiet_begin() {
  fflush(stderr); fflush(stdout); cerr.flush(); cout.flush();
  if (getenv("REF_DRIVER_VERBOSE")) return;
  S.saved_stderr = dup(2);
  int nul = open("/dev/null", O_WRONLY);
  dup2(nul, 2);
  close(nul);
}
void quiet_end() {
  cerr.flush(); fflush(stderr);
  if (S.saved_stderr >= 0) { dup2(S.saved_stderr, 2); close(S.saved_stderr); S.saved_stderr = -1; }
}

void fill_call(const cnv_st& c, ref_call* o) {
  o->start = c.start; o->end = c.end; o->type = c.type; o->geno = c.geno; o->status = c.status;
  o->length = c.length; o->pad = 0;
  double q1 = c.p1 < 1.0E-10 ? 99 : -10.0 * log(c.p1) / log(10.0);   // as printed by cnv_format1
  o->qscore = (int)q1;
  o->score = c.score; o->p1 = c.p1; o->cnvmed = c.cnvmed; o->cnvsd = c.cnvsd; o->cnviqr = c.cnviqr;
  o->refmed = c.refmed; o->refsd = c.refsd; o->refiqr = c.refiqr;
}

double now_s() {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

bool same_segs(const vector<cnv_st>& a, const vector<cnv_st>& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); ++i)
    if (a[i].start != b[i].start || a[i].end != b[i].end || a[i].type != b[i].type ||
        a[i].score != b[i].score) return false;
  return true;
}
bool same_arr(Array<int>& a, Array<int>& b) {
  if (a.size() != b.size()) return false;
  for (int i = 0; i < a.size(); ++i) if (a[i] != b[i]) return false;
  return true;
}

}  // namespace

extern "C" {

void ref_default_params(ref_params* p) {   // rsi.cpp:34-98 defaults
  p->m = 101; p->gcadjust = 1; p->trans = 0; p->merge = 1; p->maxchkbp = 100000; p->debug = 0;
  p->cap = 4.0; p->epsilon = 1.5; p->threshold = -1.0; p->chklen = 2.5; p->minmlen = 3.01;
  p->buffer = 0.05; p->p = 0.05;
}

// Set the reference's globals and load one chromosome: what load_data_from_text does between
// read_fasta and the parse loop (loaddata.cpp:478-492), with the depth array handed in directly.
int ref_load(const ref_params* p, const int32_t* depth, const char* fasta, int32_t n,
             const char* chrname, const char* logpath) {
  rsi::m = p->m; rsi::gcadjust = p->gcadjust != 0; rsi::merge = p->merge != 0;
  rsi::trans = p->trans == 0 ? "NBN" : (p->trans == 1 ? "MED" : "ALL");
  rsi::maxchkbp = p->maxchkbp; rsi::debug = p->debug != 0; rsi::cap = p->cap;
  rsi::epsilon = p->epsilon; rsi::threshold = p->threshold; rsi::chklen = p->chklen;
  rsi::minmlen = p->minmlen; rsi::buffer = p->buffer; rsi::p = p->p;
  rsi::plot = false;
  rsi::chr = chrname ? chrname : "chrS";
  rsi::target_name.clear();
  rsi::target_name.push_back(rsi::chr);
  rsi::tid = 0;
  rsi::rdfile = "synthetic"; rsi::bamfile = "";
  if (rsi::fout.is_open()) rsi::fout.close();
  rsi::fout.clear();
  rsi::fout.open(logpath && logpath[0] ? logpath : "/dev/null");

  quiet_begin();
  string FASTA(fasta, (size_t)n);
  S.GC.resize(0);
  S.GC.resize(n);
  S.GC.assign(false);
  for (int k = 0; k < n; ++k) S.GC[k] = (FASTA[k] == 'G' || FASTA[k] == 'C');   // loaddata.cpp:481-483
  get_noseq_regions(FASTA);
  S.RD.resize(0);   // Array::resize keeps a shrunken logical size when the capacity already matches
  S.RD.resize(n);
  for (int k = 0; k < n; ++k) S.RD[k] = depth[k];
  rsi::start = 1;
  rsi::end = S.RD.size();
  quiet_end();
  return (int)rsi::noncodelist.size();
}

int ref_stage_gc(void)  { quiet_begin(); if (rsi::gcadjust) checkgccontent(S.RD, S.GC); quiet_end(); return S.RD.size(); }
int ref_stage_cap(void) { quiet_begin(); if (rsi::cap > 1) apply_cap(S.RD); quiet_end(); return S.RD.size(); }
int ref_stage_concat(void) {   // rsi.cpp:2200-2203
  quiet_begin();
  concatenate_data(S.RD);
  rsi::RDmedian = _median(&S.RD[0], S.RD.size());
  rsi::RDsd = sqrt(variance(S.RD, 0, S.RD.size() - 1, 0.0, -1));
  quiet_end();
  return S.RD.size();
}
int ref_rd_size(void) { return S.RD.size(); }
int ref_get_rd(int32_t* out, int32_t cap) {
  int n = S.RD.size() < cap ? S.RD.size() : cap;
  for (int i = 0; i < n; ++i) out[i] = S.RD[i];
  return n;
}
int ref_get_noncode(int32_t* beg, int32_t* end, int32_t cap) {
  int n = (int)rsi::noncodelist.size();
  for (int i = 0; i < n && i < cap; ++i) { beg[i] = rsi::noncodelist[i].start; end[i] = rsi::noncodelist[i].end; }
  return n;
}
void ref_get_chrom_scalars(double* out) { out[0] = rsi::RDmedian; out[1] = rsi::RDsd; }

// median_transfer + RDmedint + negative_binomial_transfer exactly as detectcnv calls them
// (rsi.cpp:1816-1826).
int ref_stage_bins(void) {
  quiet_begin();
  int nb = S.RD.size() / rsi::m;
  S.RDmed.resize(nb); S.RDmedint.resize(nb); S.RDnbn.resize(nb);
  median_transfer(S.RD, rsi::m, S.RDmed);
  for (int i = 0; i < S.RDmed.size(); ++i) S.RDmedint[i] = (int)(S.RDmed[i] + 0.5);
  rsi::RDmedian = _median(&S.RD[0], S.RD.size());
  negative_binomial_transfer(S.RD, rsi::m, S.RDnbn);
  rsi::factor = sqrt(2.0 * (1.0 + rsi::epsilon) * log(3.1E9));
  rsi::Lmax = 10000 / rsi::m;
  if (rsi::Lmax < 20) rsi::Lmax = 20;
  quiet_end();
  return nb;
}
int ref_get_bins(float* rdmed, int32_t* rdmedint, float* rdnbn, int32_t cap) {
  int nb = S.RDmed.size();
  for (int i = 0; i < nb && i < cap; ++i) {
    if (rdmed) rdmed[i] = S.RDmed[i];
    if (rdmedint) rdmedint[i] = S.RDmedint[i];
    if (rdnbn) rdnbn[i] = S.RDnbn[i];
  }
  return nb;
}

// The scan with its intermediates.  use_med=0: the rsicnvnbn sequence; use_med=1: rsicnvmed.
int ref_scan_stepwise(int use_med, ref_scan_scalars* sc) {
  quiet_begin();
  Array<float>& T = use_med ? S.RDmed : S.RDnbn;
  int nb = T.size();
  Array<float> tmp(nb);
  double tmedian, tsigma, tlamda, target, dev;
  int Lmax = rsi::Lmax, cal_max;
  if (!use_med) {                                   // rsi.cpp:1273-1289
    tmedian = _median(&T[0], T.size());
    for (int i = 0; i < nb; ++i) tmp[i] = abs(T[i] - tmedian);
    tsigma = _median(&tmp[0], tmp.size()) / 0.6745;
    tlamda = rsi::factor * tsigma;
    target = (T[2] - T[0]) * sqrt(2.5);
    tlamda = max(tlamda, target);
    double dnb = abs(T[2] - T[0]) + 0.0001;
    cal_max = pow(tlamda * 2 / dnb, 2);
    dev = tsigma * 3.0;
  } else {                                          // rsi.cpp:1413-1433, 1454
    tmedian = rsi::RDmedian;
    for (int i = 0; i < nb; ++i) tmp[i] = abs(T[i] - tmedian);
    tsigma = _median(&tmp[0], tmp.size()) / 0.6745;
    tlamda = rsi::factor * tsigma;
    target = tmedian * sqrt(2.0);
    tlamda = max(tlamda, target);
    if (rsi::threshold > 0) tlamda = tmedian * rsi::threshold;
    cal_max = pow(tlamda * 4 / (tmedian + 0.001), 2);
    dev = tmedian * 0.6;
  }
  if (Lmax < cal_max) Lmax = cal_max;
  sc->tmedian1 = tmedian; sc->tsigma1 = tsigma; sc->tlamda1 = tlamda; sc->target_tlamda = target;
  sc->Lmax = Lmax; sc->cal_max = cal_max;

  Array<int> status(nb, 0);
  rsistatus(T, S.RDmedint, tmedian, tlamda, Lmax, status);
  S.st_pass1 = status;
  filterstatus(T, dev, status);
  S.st_filtered = status;
  int k = 0;
  for (int i = 0; i < nb; ++i) if (status[i] == 0) { tmp[k] = T[i]; k++; }
  if (k > nb / 2) {
    tmedian = _median(&tmp[0], k);
    for (int i = 0; i < k; ++i) tmp[i] = abs(tmp[i] - tmedian);
    tsigma = _median(&tmp[0], k) / 0.6745;
    tlamda = rsi::factor * tsigma;
    tlamda = max(tlamda, target);
  }
  sc->tmedian2 = tmedian; sc->tsigma2 = tsigma; sc->tlamda2 = tlamda;
  rsistatus(T, S.RDmedint, tmedian, tlamda, Lmax, status);
  S.st_final = status;
  vector<cnv_st> segs, kept;
  get_rsi_segments(T, status, tmedian, segs);
  for (size_t i = 0; i < segs.size(); ++i) {
    if (abs(segs[i].score) < tlamda * 0.5) segs[i].status = -9;
    if (segs[i].status != -9) kept.push_back(segs[i]);
  }
  S.seglist = kept;
  sc->nseg = (int)kept.size();

  // self-check against the reference's own driver function
  Array<int> status_ref(nb, 0);
  vector<cnv_st> segs_ref;
  double keep_median = rsi::RDmedian;
  if (!use_med) rsicnvnbn(T, S.RDmedint, status_ref, segs_ref);
  else rsicnvmed(T, S.RDmedint, status_ref, segs_ref);
  rsi::RDmedian = keep_median;
  sc->stepwise_matches_reference = (same_arr(status_ref, S.st_final) && same_segs(segs_ref, kept)) ? 1 : 0;
  quiet_end();
  return nb;
}
int ref_get_status(int which, int32_t* out, int32_t cap) {
  Array<int>& A = which == 0 ? S.st_pass1 : (which == 1 ? S.st_filtered : S.st_final);
  for (int i = 0; i < A.size() && i < cap; ++i) out[i] = A[i];
  return A.size();
}
int ref_get_segs(ref_call* out, int32_t cap) {
  for (size_t i = 0; i < S.seglist.size() && (int)i < cap; ++i) fill_call(S.seglist[i], &out[i]);
  return (int)S.seglist.size();
}
// areblockscnv on the stepwise segments (bin space), rsi.cpp:1847
int ref_stage_blocks(ref_call* out, int32_t cap) {
  quiet_begin();
  vector<cnv_st> segs = S.seglist;
  areblockscnv(S.RDmedint, S.st_final, segs);
  quiet_end();
  for (size_t i = 0; i < segs.size() && (int)i < cap; ++i) fill_call(segs[i], &out[i]);
  return (int)segs.size();
}

// The real thing: detectcnv + sd_filters on the compacted array (rsi.cpp:2205-2208).
int ref_stage_detect(void) {
  quiet_begin();
  S.calls_raw.clear();
  detectcnv(S.RD, S.calls_raw);
  S.calls = S.calls_raw;
  sd_filters(S.calls);
  quiet_end();
  return (int)S.calls.size();
}
int ref_get_calls(int filtered, ref_call* out, int32_t cap) {
  vector<cnv_st>& L = filtered ? S.calls : S.calls_raw;
  for (size_t i = 0; i < L.size() && (int)i < cap; ++i) fill_call(L[i], &out[i]);
  return (int)L.size();
}
// Output rows exactly as write_cnv_to_file would print them (cnv_format1, rsi.cpp:581).
int ref_format_calls(char* buf, int32_t cap) {
  string s;
  for (size_t i = 0; i < S.calls.size(); ++i) s += cnv_format1(S.calls[i]) + "\n";
  if ((int)s.size() + 1 > cap) return -(int)s.size() - 1;
  memcpy(buf, s.c_str(), s.size() + 1);
  return (int)s.size();
}

// Whole compute-only path in one go with per-stage wall times (seconds), for bench.py's
// cpu_baseline ("kind": "reference").  stage_s[0..4] = load(GC mask+N regions), gc, cap,
// concat+median+sd, detectcnv+sd_filters.
int ref_run_timed(const ref_params* p, const int32_t* depth, const char* fasta, int32_t n, double* stage_s) {
  double t0 = now_s();
  ref_load(p, depth, fasta, n, "chrS", 0);
  double t1 = now_s();
  ref_stage_gc();
  double t2 = now_s();
  ref_stage_cap();
  double t3 = now_s();
  ref_stage_concat();
  double t4 = now_s();
  int nc = ref_stage_detect();
  double t5 = now_s();
  stage_s[0] = t1 - t0; stage_s[1] = t2 - t1; stage_s[2] = t3 - t2; stage_s[3] = t4 - t3; stage_s[4] = t5 - t4;
  return nc;
}

// plot_icnv (plotcnv.cpp:246-610) on an explicit per-base array, set up as plot_cnv does (plotcnv.cpp:613-628: plot::RD,
// plot::RDmed = _median of the array).  plot_icnv deletes its data and script file when it is done (plotcnv.cpp:606-607):
// the driver gives each a second name (a hard link) beforehand, so the bytes the reference wrote stay readable as keep_dat /
// keep_gp once the reference has removed its own names.  Nothing is re-implemented; the `gnuplot < script` the reference
// starts fails harmlessly where gnuplot is absent.  out[0] = plot::RDmed, out[1] = the reference's gnuplot_version().
int ref_plot_icnv(const ref_params* p, const int32_t* rd, int32_t n, const ref_call* c, const char* chrname, const char* title,
                  const char* format, const char* datfile, const char* gpfile, const char* imgfile, const char* keep_dat,
                  const char* keep_gp, double* out) {
  rsi::m = p->m; rsi::minmlen = p->minmlen; rsi::chklen = p->chklen;
  rsi::chr = chrname;
  rsi::target_name.clear();
  rsi::target_name.push_back(rsi::chr);
  rsi::tid = 0;
  static Array<int> RD(1);
  RD.resize(0);
  RD.resize(n);
  for (int k = 0; k < n; ++k) RD[k] = rd[k];
  plot::RD = &RD;
  plot::RDmed = _median(&RD[0], RD.size());
  plot::format = format;
  cnv_st icnv;
  icnv.tid = 0; icnv.start = c->start; icnv.end = c->end; icnv.type = c->type; icnv.length = c->length; icnv.p1 = c->p1;
  const char* names[2][2] = {{datfile, keep_dat}, {gpfile, keep_gp}};
  for (int f = 0; f < 2; ++f) {
    unlink(names[f][0]); unlink(names[f][1]);
    int fd = open(names[f][0], O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return -1;
    close(fd);
    if (link(names[f][0], names[f][1]) != 0) return -2;
  }
  quiet_begin();
  out[1] = gnuplot_version();
  plot_icnv(icnv, title, datfile, gpfile, imgfile);
  quiet_end();
  out[0] = plot::RDmed;
  return 0;
}

// Direct probes of the numeric utilities (used to pin the oracle's restatements).
double ref_median_i32(const int32_t* x, int64_t n) { return _median(const_cast<int*>(x), (size_t)n); }
double ref_median_f32(const float* x, int64_t n) { return _median(const_cast<float*>(x), (size_t)n); }
double ref_median_f64(const double* x, int64_t n) { return _median(const_cast<double*>(x), (size_t)n); }
double ref_iqr_i32(const int32_t* x, int64_t n) { return _interquartilerange(const_cast<int*>(x), (size_t)n); }
double ref_iqr_f32(const float* x, int64_t n) { return _interquartilerange(const_cast<float*>(x), (size_t)n); }
double ref_exact_median_i32(const int32_t* x, int64_t n) { return alglib::median(const_cast<int*>(x), (size_t)n); }
double ref_pnorm(double x) { return alglib::pnorm(x); }
void ref_runmean_f32(const float* y, float* smo, int32_t n, int32_t band) {
  Array<float> Y(n, y), Sm(n, 0.0f);
  runmean(Y, Sm, n, band, 1);
  for (int i = 0; i < n; ++i) smo[i] = Sm[i];
}

}  // extern "C"
