// plot_host.cpp -- the data (.dat) and script (.gp) files of a call's gnuplot figure, as plot_icnv writes them
// (plotcnv.cpp:245-610), behind the command line's -p.  SURVEY.md section 8f row 4: presentation, no GPU work -- the
// per-base array it reads is the capped, GC-adjusted depth the device path leaves (rsi_hot_fetch "rd_concat"), expanded by
// the removed N regions (expand_data, loaddata.cpp:140-183).  The reference runs gnuplot on the pair and deletes both
// files; this writer only produces them (the command line runs gnuplot when there is one, and keeps the files when not).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rsi_hot.h"
#include "hostmath.h"

namespace {

constexpr int kPlotPoints = 30000;   // plot::pts, plotcnv.cpp:29

// runmeantp with end_rule = 1 (wufunctions.cpp:573-647): centred window of `band` values, double sliding sum, the ends filled
// with the first / last mean; the float overload rounds every mean to float (wufunctions.cpp:640-646)
void running_mean(const std::vector<int>& y, std::vector<float>& smo, int band) {
  const int n = (int)y.size();
  std::vector<double> s((size_t)n, 0.0);
  double sum = 0;
  for (int i = 0; i < band; ++i) sum += (double)y[(size_t)i];
  double mean = sum / double(band);
  int band2 = band / 2;
  for (int i = 0; i < band2; ++i) s[(size_t)i] = mean;
  s[(size_t)band2] = mean;
  ++band2;
  int ismo = band2;
  for (int first = 1, last = band; last < n; ++first, ++last, ++ismo) {
    sum = sum - (double)y[(size_t)first - 1] + (double)y[(size_t)last];
    mean = sum / double(band);
    s[(size_t)ismo] = mean;
  }
  for (int i = ismo; i < n; ++i) s[(size_t)i] = mean;
  smo.resize((size_t)n);
  for (int i = 0; i < n; ++i) smo[(size_t)i] = (float)s[(size_t)i];
}

const char* kType[3] = {"DEL", "DUP", "UNKNOWN"};

}  // namespace

extern "C" {

// Writes <base>.dat and <base>.gp for call `c` (coordinates in the expanded array, as printed in the output table).
// rd: the expanded per-base depth (n values); chrom_median: _median of that array (plot::RDmed, plotcnv.cpp:625);
// m / minmlen / chklen: the run's parameters (rsi.cpp:34-98); format: "ps", "eps" or "png" (plot::format);
// gnuplot_version: what `gnuplot -V` reports, -1 = none found (gnuplot_version(), plotcnv.cpp:51-65).  The script has two
// dialects and the reference picks the newer one only above 4.19 (plotcnv.cpp:512) -- "none found" gets the older one.
// Returns RSI_OK, or RSI_ERR_BAD_ARG when the call lies outside the array (the reference draws an empty frame there).
int rsi_plot_write_files(const rsi_call* c, const char* title_in, const int32_t* rd, int64_t n, double chrom_median, int m,
                         double minmlen, double chklen, const char* format, double gnuplot_version, const char* datfile,
                         const char* gpfile, const char* imgfile) {
  if (!c || !title_in || !rd || n <= 0 || !format || !datfile || !gpfile || !imgfile) return RSI_ERR_BAD_ARG;
  int cs = c->start, ce = c->end;
  if (cs > ce) std::swap(cs, ce);
  const int size = (int)n;
  if (cs > size || ce > size) return RSI_ERR_BAD_ARG;
  std::string term = "png xffffff x222222";
  const std::string fmt = format;
  if (fmt == "ps") term = "postscript color enhanced solid";
  if (fmt == "eps") term = "postscript eps enhanced solid";
  int d = ce - cs + 1;
  if (d < m * minmlen) d = (int)(m * minmlen);
  int i1 = (int)(cs - chklen * d), i2 = (int)(ce + chklen * d);
  if (i1 < 1) i1 = 1;
  if (i1 > size - 1) i1 = size - 1;
  if (i2 > size - 1) i2 = size - 1;
  int c1 = cs - 1, c2 = ce - 1;
  if (c1 < 0) c1 = 0;
  if (c1 >= size) c1 = size - 1;
  if (c2 >= size) c2 = size - 1;
  // the neighbourhood: bases i1 .. start-1 and end+1 .. i2 (1-based positions: value at index position - 1)
  std::vector<int> ref((size_t)abs(cs - i1 + i2 - ce));
  int k = 0;
  double sum = 0;
  for (int i = i1; i < cs; ++i, ++k) {
    const int ic = i - 1;
    if (ic >= size) break;
    ref[(size_t)k] = ic < 0 ? 0 : rd[ic];
    sum += ref[(size_t)k];
  }
  for (int i = ce + 1; i <= i2; ++i, ++k) {
    const int ic = i - 1;
    if (ic >= size) break;
    ref[(size_t)k] = ic < 0 ? 0 : rd[ic];
    sum += ref[(size_t)k];
  }
  if (k != (int)ref.size() || ref.empty()) return RSI_ERR_BAD_ARG;   // "size error": the reference returns without a figure
  const double refmean = sum / double(k);
  const int nbody = c2 - c1 + 1;
  if (nbody <= 0) return RSI_ERR_BAD_ARG;
  const rsih::Quantiles qc = rsih::grid_quantiles(rd + c1, (size_t)nbody);   // _median / _lowerquartile / _upperquartile
  std::vector<float> runmean;
  d = abs(ce - cs) + 1;
  int band = d + ((d + 1) % 2) * 1;
  if (band > (int)ref.size()) band = (int)ref.size() / 2 + (((int)ref.size() / 2 + 1) % 2) * 1;
  while (band > (int)ref.size()) band -= 2;
  if (band <= 0) return RSI_ERR_BAD_ARG;
  running_mean(ref, runmean, band);
  const rsih::Quantiles qr = rsih::grid_quantiles(ref.data(), ref.size());
  int ymax = (int)(qr.med * 2.25);
  const double y2max = (double)ymax / refmean / 2.0;
  if (qc.med < qr.med) ymax = (int)(qr.med * 2);
  else ymax = (int)(qc.uqt + 1.5 * (qr.uqt - qr.lqt));

  std::ofstream D(datfile);
  if (!D) return RSI_ERR_BAD_ARG;
  D << "#" << cs << " ~ " << ce << "  " << c->length << "  " << kType[c->type < 0 || c->type > 2 ? 2 : c->type] << "  " << c->p1 << std::endl;
  int RDmax = 0;
  double step = (double)(i2 - i1 + 1) / (double)kPlotPoints;
  if (step < 1.0) step = 1.0;
  int i = 0;
  // block 0: before the call: position, depth, running mean of the neighbourhood
  for (double ir = (double)i1 + 0.00001; ir < (double)cs + 0.000011; ir += step) {
    i = (int)ir;
    const int ic = i - 1;
    if (ic < 0 || ic >= size) { D << i << "\t0\tNaN" << std::endl; continue; }
    D << i << "\t" << rd[ic] << "\t" << runmean[(size_t)(i - i1)] << std::endl;
    if (ic > 0 && rd[ic] > RDmax) RDmax = rd[ic];
  }
  D << std::endl << std::endl;
  // block 1: the call: position, depth
  for (double ir = (double)cs + 0.00001; ir <= (double)ce + 0.000011; ir += step) {
    const int ic = (int)(ir - 1);
    if (ic < 0 || ic >= size) { D << i << "\t0\tNaN" << std::endl; continue; }
    D << (int)ir << "\t" << rd[ic] << "\t" << "NaN" << std::endl;
    if (ic > 0 && rd[ic] > RDmax) RDmax = rd[ic];
  }
  D << std::endl << std::endl;
  // block 2: behind the call
  for (double ir = (double)ce + 1.00001; ir <= (double)i2 + 0.000011; ir += step) {
    const int ic = (int)(ir - 1);
    if (ic < 0 || ic >= size) { D << i << "\t0\tNaN" << std::endl; continue; }
    const int ri = (int)(ir - i1 - d);
    D << (int)ir << "\t" << rd[ic] << "\t" << runmean[(size_t)std::min(std::max(ri, 0), (int)runmean.size() - 1)] << std::endl;
    if (ic > 0 && rd[ic] > RDmax) RDmax = rd[ic];
  }
  D << std::endl << std::endl;
  // block 3: the line that joins the two running means
  int imid1 = cs - 3 - i1, imid2 = ce + 1 - i1 - d;
  if (imid1 < 0) imid1 = 0;
  if (imid2 >= (int)runmean.size()) imid2 = (int)runmean.size() - 1;
  if (imid2 < 0) imid2 = 0;
  D << cs << "\t" << runmean[(size_t)imid1] << std::endl << ce << "\t" << runmean[(size_t)imid2] << std::endl;
  D << std::endl << std::endl;
  // blocks 4-6: median and quartiles of the call; 7-9: of the neighbourhood; 10: nothing
  const double cq[3] = {qc.med, qc.lqt, qc.uqt}, rq[3] = {qr.med, qr.lqt, qr.uqt};
  for (int b = 0; b < 3; ++b) D << cs << "\t" << cq[b] << std::endl << ce << "\t" << cq[b] << std::endl << std::endl << std::endl;
  for (int b = 0; b < 3; ++b) D << i1 << "\t" << rq[b] << std::endl << i2 << "\t" << rq[b] << std::endl << std::endl << std::endl;
  D << "NaN\tNaN\n" << "NaN\tNaN\n" << std::endl << std::endl;
  D.close();

  std::string title = title_in;
  std::replace(title.begin(), title.end(), '~', '-');
  if (ymax > RDmax) ymax = RDmax + RDmax / 10;
  d = (i2 - i1 + 1) / 6;
  i1 = i1 + d / 2;
  i2 = i2 - d / 2;
  std::ofstream G(gpfile);
  if (!G) return RSI_ERR_BAD_ARG;
  if (gnuplot_version > 4.19) {
    G << "f=\"" << datfile << "\"" << std::endl
      << "set datafile missing 'NaN'" << std::endl
      << "info=\"" << title << " \"" << std::endl
      << "set terminal " << term << std::endl
      << "set output \"" << imgfile << "\"" << std::endl
      << "#set nokey" << std::endl
      << "##set label 1 info at graph  0.25, graph  0.9" << std::endl
      << "set title info offset 0,-0.5" << std::endl
      << "set xrange [" << i1 << ":" << i2 << "]" << std::endl
      << "set xtics " << i1 << "," << d << "," << i2 << std::endl
      << "set yrange [0:" << ymax << "]" << std::endl;
    G << "set y2range[0:" << y2max << "]" << std::endl;
    G << "set ytics nomirror" << std::endl;
    G << "plot \\" << std::endl
      << "f in 0 u 1:2 w p pt 7 ps 0.5 lt rgb \"blue\" t \"Neighbor\", \\" << std::endl
      << "f in 1 u 1:2 w p pt 7 ps 0.5 lt rgb \"red\" t \"CNV\", \\" << std::endl
      << "f in 2 u 1:2 w p pt 7 ps 0.5 lt rgb \"blue\" not, \\" << std::endl
      << "f in 0 u 1:3 w l lt 1 lw 8 lc rgb \"green\" t \"Runmean\", \\" << std::endl
      << "f in 2 u 1:3 w l lt 1 lw 8 lc rgb \"green\" not, \\" << std::endl
      << "f in 3 u 1:2 w l lt 0 lw 8 lc rgb \"green\" not, \\" << std::endl
      << "f in 4 u 1:2 w l lt 1 lw 8 lc rgb \"cyan\" t \"CNV med\", \\" << std::endl
      << "f in 5 u 1:2 w l lt 0 lw 8 lc rgb \"cyan\" t \"CNV 1st,3rd quart\", \\" << std::endl
      << "f in 6 u 1:2 w l lt 0 lw 8 lc rgb \"cyan\" not, \\" << std::endl
      << qr.lqt << " w l lt 0 lw 8 lc rgb \"green\" t \"Neighbor 1st,3rd quar\", \\" << std::endl
      << qr.uqt << " w l lt 0 lw 8 lc rgb \"green\" not, \\" << std::endl;
    if (chrom_median > 0) G << chrom_median << " w l lt 4 lw 4 t \"CHROM med\", \\" << std::endl;
    G << "f in 10 u 1:2 not" << std::endl;
  } else {   // gnuplot up to 4.1: no string variables, no rgb colours
    G << "#f=\"" << datfile << "\"" << std::endl
      << "#info=\"" << title << " \"" << std::endl
      << "set terminal " << term << std::endl
      << "set output \"" << imgfile << "\"" << std::endl
      << "#set nokey" << std::endl
      << "set title \"" << title << "\"" << " 0,-0.5" << std::endl
      << "set xrange [" << i1 << ":" << i2 << "]" << std::endl
      << "set xtics " << i1 << "," << d << "," << i2 << std::endl
      << "set yrange [0:" << ymax << "]" << std::endl;
    G << "set ytics nomirror" << std::endl;
    G << "set y2range[0:" << y2max << "]" << std::endl;
    G << "plot \\" << std::endl
      << "\"" << datfile << "\" in 0 u 1:2 w p pt 7 ps 0.5 lt 3 not, \\" << std::endl
      << "\"" << datfile << "\" in 1 u 1:2 w p pt 7 ps 0.5 lt 1 not, \\" << std::endl
      << "\"" << datfile << "\" in 2 u 1:2 w p pt 7 ps 0.5 lt 3 not, \\" << std::endl
      << "\"" << datfile << "\" in 0 u 1:3 w l lt 2 lw 8 t \"runmean\", \\" << std::endl
      << "\"" << datfile << "\" in 2 u 1:3 w l lt 2 lw 8 not, \\" << std::endl
      << "\"" << datfile << "\" in 3 u 1:2 w l lt 2 lw 2 not, \\" << std::endl
      << "\"" << datfile << "\" in 4 u 1:2 w l lt 5 lw 8 t \"CNV med\", \\" << std::endl
      << "\"" << datfile << "\" in 5 u 1:2 w l lt 5 lw 4 t \"CNV 1st,3rd quart\", \\" << std::endl
      << "\"" << datfile << "\" in 6 u 1:2 w l lt 5 lw 4 not, \\" << std::endl
      << qr.lqt << " w l lt 2 lw 2 t \"Neighbor 1st,3rd quart\", \\" << std::endl
      << qr.uqt << " w l lt 2 lw 2 not , \\" << std::endl;
    if (chrom_median > 0) G << chrom_median << " w l lt 4 lw 4 t \"WG mean\", \\" << std::endl;
    G << "\"" << datfile << "\" in 10 u 1:2 not" << std::endl;
  }
  G << "set output" << std::endl << "quit" << std::endl;
  return RSI_OK;
}

// expand_data (loaddata.cpp:140-183): the compacted array with the removed regions back in, as zeros.  regions: npairs
// (start, end) pairs in reference coordinates, ascending; out: n values.
int rsi_plot_expand(const int32_t* rdc, int64_t ncompact, const int32_t* regions, int npairs, int32_t* out, int64_t n) {
  if (!rdc || !out || ncompact < 0 || n < ncompact || (npairs > 0 && !regions)) return RSI_ERR_BAD_ARG;
  int64_t removed = 0;
  for (int k = 0; k < npairs; ++k) {
    if (regions[2 * k] < 0 || regions[2 * k + 1] < regions[2 * k] || regions[2 * k + 1] >= n || (k > 0 && regions[2 * k] <= regions[2 * k - 1])) return RSI_ERR_BAD_ARG;
    removed += (int64_t)regions[2 * k + 1] - regions[2 * k] + 1;
  }
  if (ncompact + removed != n) return RSI_ERR_BAD_ARG;   // "cannot expand RD array", loaddata.cpp:149-153
  int64_t src = 0, dst = 0;
  for (int k = 0; k < npairs; ++k) {
    const int64_t s = regions[2 * k], e = regions[2 * k + 1];
    while (dst < s && src < ncompact && dst < n) out[dst++] = rdc[src++];
    while (dst <= e && dst < n) out[dst++] = 0;
  }
  while (src < ncompact && dst < n) out[dst++] = rdc[src++];
  return (src == ncompact && dst == n) ? RSI_OK : RSI_ERR_BAD_ARG;
}

}  // extern "C"
