// cli.cpp -- `rsicnv rsi ...`: the reference's command line (rsi.cpp:1949-2068, 2069-2217) in
// front of librsi_hot.so.  Same flags and defaults, same output file (header lines, columns,
// number formatting).  Inputs: a depth file (-d RDFILE -c RNAME, parsed on the device) or a BAM
// file (-b BAMFILE [-c RNAME], piled up on the device, calls annotated with RP / Q0 from its read
// pairs); plot, stat and pin are outside the accelerated path and say so.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <thread>
#include <algorithm>

#include "../../include/rsi_hot.h"
#include "hostmath.h"

namespace {

struct Options {
  std::string function = "rsi", rdfile, bamfile, reffile, outfile = "rsiout.txt", chr = "1-22XY", plotfolder = "cnv_plots";
  rsi_params P;
  int minq = 0, min_baseQ = 13, device = 0;
  int gpus = 0;      // -gpus N: chromosomes spread over N devices, several in flight per device (0: one context, one at a time)
  int workers = 4;   // -workers W: chromosomes in flight per device with -gpus
  bool saverd = false, plot = true, plotfiles = false;
};

int usage() {
  std::cerr << "Usage:\n\n1. detect CNV\n\n"
            << "   rsicnv rsi <options> [-b BAMFILE | -d RDFILE -c RNAME ] -f REFFILE \n"
            << "\nOptions:\n"
            << "   -m   INT  bin size, default=101\n"
            << "   -q   INT  minimum mapping quality, default=0\n"
            << "   -Q   INT  minimum base quality, default=10\n"
            << "   -cap INT  cap read depth at INT*median, dafault=4\n"
            << "             if INT<0, do not cap read depth\n"
            << "   -NOGC     do not adjust GC content, default=adjust\n"
            << "   -MED      only use median transformation \n"
            << "   -NB       only use negative binomial transformation (default)\n"
            << "   -o   STR  output file, default=rsiout.txt \n"
            << "   -np       do not plot CNV\n"
            << "   -gpu INT  HIP device to run on, default=0\n"
            << "   -gpus INT spread the chromosomes of a BAM over INT devices (longest first), several in flight per\n"
            << "             device (-workers INT, default=4); rows are written in BAM header order all the same\n"
            << "\nNote:\n"
            << "   This build runs the read-depth hot path on an MI355X; input is a read depth file\n"
            << "   (samtools mpileup BAM | cut -f2,4) with -c RNAME, or a coordinate-sorted BAM file (all\n"
            << "   chromosomes with reads, or the one named with -c), plus the indexed reference.\n"
            << "   -s saves the BAM's depth to OUT.RNAME_rd.\n"
            << std::endl;
  return 0;
}

// get_parameters, rsi.cpp:1986-2068
void parse(int argc, char** argv, Options& o) {
  rsi_default_params(&o.P);
  std::vector<std::string> a(argv, argv + argc);
  if (a.size() < 2) exit(usage());
  size_t i = 1;
  if (a[1][0] != '-') {
    o.function = a[1];
    if (o.function != "rsi" && o.function != "plot" && o.function != "stat" && o.function != "pin") {
      std::cerr << "no such function " << o.function << std::endl;
      exit(usage());
    }
    i = 2;
  }
  auto need = [&](size_t k) { if (k + 1 >= a.size()) exit(usage()); return a[k + 1]; };
  for (; i < a.size(); ++i) {
    const std::string& s = a[i];
    if (s == "-d") { o.rdfile = need(i); ++i; }
    else if (s == "-b") { o.bamfile = need(i); ++i; }
    else if (s == "-f") { o.reffile = need(i); ++i; }
    else if (s == "-v") { need(i); ++i; }
    else if (s == "-o") { o.outfile = need(i); ++i; }
    else if (s == "-c") { o.chr = need(i); ++i; }
    else if (s == "-s") o.saverd = true;
    else if (s == "-m") { o.P.m = atoi(need(i).c_str()); ++i; }
    else if (s == "-q") { o.minq = atoi(need(i).c_str()); ++i; }
    else if (s == "-Q") { o.min_baseQ = atoi(need(i).c_str()); ++i; }
    else if (s == "-L") { need(i); ++i; }
    else if (s == "-p") { o.plotfolder = need(i); ++i; }
    else if (s == "-np") o.plot = false;
    else if (s == "-plotfiles") o.plotfiles = true;
    else if (s == "-threshold") { o.P.threshold = atof(need(i).c_str()); ++i; }
    else if (s == "-e") { o.P.epsilon = atof(need(i).c_str()); ++i; }
    else if (s == "-cap") { o.P.cap = atof(need(i).c_str()); ++i; }
    else if (s == "-reflen") { o.P.chklen = atof(need(i).c_str()); ++i; }
    else if (s == "-maxchkbp") { o.P.maxchkbp = atoi(need(i).c_str()); ++i; }
    else if (s == "-debug") o.P.debug = 1;
    else if (s == "-MED") o.P.trans = 1;
    else if (s == "-NB") o.P.trans = 0;
    else if (s == "-ALL") o.P.trans = 2;
    else if (s == "-nomerge") o.P.merge = 0;
    else if (s == "-hist" || s == "-overlap" || s == "-combine" || s == "-nocode") {}
    else if (s == "-NOGC") o.P.gcadjust = 0;
    else if (s == "-gpu") { o.device = atoi(need(i).c_str()); ++i; }
    else if (s == "-gpus") { o.gpus = atoi(need(i).c_str()); ++i; }
    else if (s == "-workers") { o.workers = atoi(need(i).c_str()); ++i; }
    else { std::cerr << "unknown option " << s << std::endl; exit(usage()); }
  }
  if (o.rdfile.empty() && o.bamfile.empty()) { std::cerr << "need input file " << std::endl; exit(usage()); }
  if (o.reffile.empty() && o.function == "rsi") { std::cerr << "need reference file " << std::endl; exit(usage()); }
  if (o.outfile == o.bamfile || o.outfile == o.rdfile) { std::cerr << "output file is same as input file " << std::endl; exit(usage()); }
  if (!o.rdfile.empty() && o.chr.empty()) { std::cerr << "readdepth file and chromosome must be specified together" << std::endl; exit(usage()); }
  if ((o.P.m % 2) != 1) { o.P.m += 1; std::cerr << "m is changed to " << o.P.m << std::endl; }   // rsi.cpp:2061-2064
}

// read_fasta, readref.cpp:10-86: one chromosome through the .fai index
bool read_fasta(const std::string& fasta, const std::string& chr, std::string& ref) {
  std::ifstream fai((fasta + ".fai").c_str());
  if (!fai) { std::cerr << "[read_fasta] Index file " << fasta << ".fai not found\n"; return false; }
  std::string name, line;
  long len = 0, offset = 0, nbases = 0, lwidth = 0;
  bool found = false;
  while (std::getline(fai, line)) {
    std::istringstream iss(line);
    iss >> name >> len >> offset >> nbases >> lwidth;
    if (name == chr || name == "chr" + chr) { found = true; break; }
  }
  if (!found) { std::cerr << chr << " not found in fai index\n"; return false; }
  const long flen = len + (len / nbases) * (lwidth - nbases);
  std::vector<char> buf((size_t)flen + 1);
  std::ifstream fin(fasta.c_str(), std::ios::binary);
  fin.seekg(offset, std::ios::beg);
  fin.read(buf.data(), flen);
  const long got = (long)fin.gcount();
  ref.resize((size_t)len);
  long k = 0;
  for (long i = 0; i < got && k < len; ++i) if (buf[i] != '\n') ref[(size_t)k++] = buf[i];
  if (k != len) { std::cerr << "Error reading the reference fasta\nread " << k << " bases\nexpecting " << len << " bases" << std::endl; return false; }
  return true;
}

const char* kHeader =
    "#CHROM\tSTART\tEND\tTYPE\tSCORE\tLENGTH\tCNV_MED(CNV_SD);NEIGHBOR_MED(NEIGHBOR_RUNMEANSD);CHR_MED(CHR_SD)\t"
    "RP=#support_read_pairs;Q0=#fraction_of_Q0_reads\tMETHOD";

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// What one iteration of the reference's chromosome loop (rsi.cpp:2189-2217) leaves behind: its log lines and its rows.
struct ChromOutput {
  std::string log;                 // everything the iteration prints (stderr + OUT.log), in order
  std::vector<std::string> rows;   // output rows (cnv_format1), without the header
  bool populated = false;          // the chromosome was processed (has reads / could be read): it counts for the header
  bool fatal = false;              // the single-chromosome modes stop here (the reference exits)
};

// One chromosome on one context: FASTA, depth (text or BAM, on the device), the hot path, RP / Q0, the rows.
void process_chromosome(rsi_ctx* ctx, const Options& o, const std::string& chr, bool many, ChromOutput& co) {
  const bool from_bam = !o.bamfile.empty();
  std::ostringstream info;
  info << "#processing " << chr << "\n";
  const double t0 = now_s();
  std::string fasta;
  if (!read_fasta(o.reffile, chr, fasta)) { co.log = info.str(); co.fatal = !many; return; }
  const double t1 = now_s();

  rsi_result* res = nullptr;
  rsi_text_stats ts;
  rsi_bam_stats bs;
  memset(&ts, 0, sizeof(ts)); memset(&bs, 0, sizeof(bs));
  // the depth comes from a text file parsed on the device (load_data_from_text's loop, loaddata.cpp:496-517) or from
  // the BAM file's reads, inflated on the host and piled up on the device (load_data_from_bam, loaddata.cpp:277-333)
  const int rc = from_bam
      ? rsi_hot_run_bam(ctx, &o.P, o.bamfile.c_str(), chr.c_str(), o.minq, o.min_baseQ, reinterpret_cast<const uint8_t*>(fasta.data()), (int64_t)fasta.size(), &res, &bs)
      : rsi_hot_run_text(ctx, &o.P, o.rdfile.c_str(), reinterpret_cast<const uint8_t*>(fasta.data()), (int64_t)fasta.size(), &res, &ts);
  if (rc != RSI_OK) {   // the reference prints its message and exits with status 0
    info << rsi_hot_last_error(ctx) << "\n";
    co.log = info.str();
    co.fatal = !(many && from_bam && bs.on_chrom == 0);   // a reference without reads is simply not "populated" (rsi.cpp:2125)
    return;
  }
  const double t2 = now_s();
  if (from_bam && many && bs.on_chrom == 0) {   // not "populated": the reference does not process it (rsi.cpp:2125)
    info << "no reads on " << chr << "\n";
    co.log = info.str();
    rsi_result_free(res);
    return;
  }
  co.populated = true;
  if (from_bam && o.saverd) {   // -s: write_rd_to_file, loaddata.cpp:340-344, 464-470
    const std::string dump = o.outfile + "." + chr + "_rd";
    std::vector<int32_t> rd((size_t)bs.n);
    rsi_hot_fetch_i32(ctx, "depth_in", rd.data(), bs.n);
    FILE* f = fopen(dump.c_str(), "w");
    if (f) { for (int64_t i = 0; i < bs.n; ++i) fprintf(f, "%lld\t%d\n", (long long)i + 1, rd[(size_t)i]); fclose(f); }
    info << "RD of " << chr << " is saved to " << dump << "\n";
  }
  const rsi_chrom_stats* S = rsi_result_stats(res);
  info << "#Noseq regions excluded\n";
  {
    std::vector<int32_t> pairs((size_t)S->n_noncode * 2 + 2);
    const int k = rsi_result_noncode(res, pairs.data(), S->n_noncode);
    for (int i = 0; i < k; ++i) info << chr << "\t" << pairs[2 * i] << "\t" << pairs[2 * i + 1] << "\n";
  }
  if (o.P.gcadjust) info << "RD mean before GC adjust = " << S->gc_rdmean << "\n";
  if (o.P.cap > 1) info << "applying cap " << o.P.cap << " times of mean " << S->cap_median << "\ncap = " << o.P.cap * S->cap_median << "\n";
  info << "region  : " << chr << ":1-" << S->n_compact << "\nmedian  : " << S->RDmedian << "\nrs::m   : " << o.P.m << "\nrs::cap : " << o.P.cap << "\n";
  {   // the scan's diagnostic lines as the reference logs them: the NB transform's median / MAD (rsi.cpp:1140-1141), the per-L
      // lines of the two rsistatus passes (rsi.cpp:1221-1224, 1251-1254), filterstatus' level table between them (rsi.cpp:991-1002)
    char line[256];
    int n = 0;
    while ((n = rsi_result_log_line(res, n, line, (int)sizeof(line))) > 0) info << line << "\n";
  }
  info << "first pass\n\tmedian of transformations : " << S->tmedian1 << "\n\tsigma : " << S->tsigma1 << "\n\tlamda : " << S->tlamda1 << "\n"
       << "second pass\n\tmedian of transformations : " << S->tmedian2 << "\n\tsigma : " << S->tsigma2 << "\n\tlamda : " << S->tlamda2 << "\n"
       << "Selected " << rsi_result_ncalls(res, 3) << " segments for testing\n"
       << "Found " << rsi_result_ncalls(res, 1) << " CNVs before sd_filters, " << rsi_result_ncalls(res, 0) << " written\n"
       << "timing: fasta " << (t1 - t0) << " s, ";
  if (from_bam)
    info << "BAM pileup " << bs.t_total_ms * 1e-3 << " s (" << bs.bytes_compressed << " bytes compressed, " << bs.records << " reads read, " << bs.used
         << " counted, inflate " << bs.t_inflate_ms * 1e-3 << " s" << (bs.indexed ? ", index used" : ", no index: scanned from the top") << ")";
  else
    info << "depth text " << ts.t_total_ms * 1e-3 << " s (" << ts.bytes << " bytes, " << ts.lines << " lines"
         << (ts.fallback ? ", host parser: positions not increasing" : "") << ")";
  info << ", whole device path " << (t2 - t1) << " s (" << S->t_device_ms << " ms on resident inputs)\n";
  if (from_bam) {   // if ( fp_in ) cnv_stat(fp_in, bamidx, cnvlist), rsi.cpp:2210
    if (rsi_result_annotate_bam(res, o.bamfile.c_str(), chr.c_str()) != RSI_OK) info << "RP / Q0 annotation failed: " << rsi_hot_last_error(nullptr) << "\n";
  }
  char row[1024];
  for (int i = 0; i < rsi_result_ncalls(res, 0); ++i) {
    rsi_result_format_row(res, i, chr.c_str(), row, (int)sizeof(row));
    co.rows.push_back(row);
  }
  // ---- plots (rsi.cpp:2213-2216: expand_data, plot_cnv): the data / script files of every written call, piped through gnuplot
  // and deleted, as the reference does -- and like the reference only when there is a gnuplot.  -plotfiles (not a reference
  // flag) writes and keeps the files without one. ----
  // gnuplot_version() (plotcnv.cpp:51-65) once per process, not once per chromosome
  static const double gv = [] {
    double v = -1.0;
    if (FILE* pp = popen("gnuplot -V 2>/dev/null | cut -d' ' -f2", "r")) { char b[64] = {0}; if (fgets(b, sizeof(b), pp) && atof(b) > 0) v = atof(b); pclose(pp); }
    return v;
  }();
  if (o.plot && rsi_result_ncalls(res, 0) > 0 && gv <= 0 && !o.plotfiles) info << "gnuplot not found\n";
  if (o.plot && rsi_result_ncalls(res, 0) > 0 && (gv > 0 || o.plotfiles)) {
    const int64_t nc = S->n_compact, nfull = S->n;
    std::vector<int32_t> rdc((size_t)nc), full((size_t)nfull), pairs((size_t)S->n_noncode * 2 + 2);
    const int np = rsi_result_noncode(res, pairs.data(), S->n_noncode);
    if (rsi_hot_fetch_i32(ctx, "rd_concat", rdc.data(), nc) == nc && rsi_plot_expand(rdc.data(), nc, pairs.data(), np, full.data(), nfull) == RSI_OK) {
      (void)!system(("mkdir -p " + o.plotfolder).c_str());
      // plot::RDmed = _median over the expanded array (plotcnv.cpp:625): zeros of the N regions included
      const double med = rsih::grid_quantiles<int>(full.data(), (size_t)nfull).med;   // partition_stat_tp's walk, its degenerate case included
      if (gv <= 0) info << "plot data and scripts are left in " << o.plotfolder << "\n";
      const rsi_call* calls = rsi_result_calls(res, 0);
      static const char* kT[3] = {"DEL", "DUP", "UNKNOWN"};
      std::vector<std::string> psfiles;
      for (int i = 0; i < rsi_result_ncalls(res, 0); ++i) {
        const rsi_call& c = calls[i];
        std::ostringstream base, title;
        base << o.plotfolder << "/rsi_" << chr << "_" << c.start << "_" << c.end << "_" << kT[c.type < 0 || c.type > 2 ? 2 : c.type];
        title << chr << ":" << c.start << "-" << c.end << " " << c.end - c.start + 1 << " " << kT[c.type < 0 || c.type > 2 ? 2 : c.type];
        const std::string dat = base.str() + ".dat", gp = base.str() + ".gp", img = base.str() + ".ps";
        info << "plotting: " << title.str() << "\n";
        if (rsi_plot_write_files(&c, title.str().c_str(), full.data(), nfull, med, o.P.m, o.P.minmlen, o.P.chklen, "ps", gv > 0 ? gv : 5.0, dat.c_str(), gp.c_str(), img.c_str()) != RSI_OK) {
          info << "CNV exceeds reference length\n";
          continue;
        }
        if (gv > 0) { (void)!system(("gnuplot < " + gp).c_str()); remove(dat.c_str()); remove(gp.c_str()); psfiles.push_back(img); }
      }
      // plot_cnv's last step (plotcnv.cpp:665-677): with ImageMagick at hand every figure also becomes a .png
      static const bool have_convert = [] {
        bool yes = false;
        if (FILE* pp = popen("convert -version 2>/dev/null | grep Image", "r")) { char b[256]; while (fgets(b, sizeof(b), pp)) if (strstr(b, "ImageMagick")) yes = true; pclose(pp); }
        return yes;
      }();
      if (have_convert) for (const std::string& ps : psfiles) {
        info << "converting: " << ps << "\n";
        (void)!system(("convert -limit thread 1 -limit area 256MB -limit disk 512MB -density 72 -rotate 90 -background white -render -antialias -flatten " +
                       ps + " " + ps.substr(0, ps.size() - 3) + ".png").c_str());
      }
    } else info << "plots skipped: the per-base depth could not be fetched\n";
  }
  co.log = info.str();
  rsi_result_free(res);
}

}  // namespace

int main(int argc, char** argv) {
  // one hardware queue per stream of a pool (the runtime's default of 4 puts unrelated chromosomes in line behind each other);
  // read when the HIP runtime initialises, so it has to be set before the first HIP call; the caller's own setting wins
  setenv("GPU_MAX_HW_QUEUES", "32", 0);
  Options o;
  parse(argc, argv, o);
  if (o.function != "rsi") {
    std::cerr << "rsicnv " << o.function << ": not part of the accelerated read-depth path in this build" << std::endl;
    return 0;
  }
  const bool from_bam = !o.bamfile.empty();
  std::ofstream log((o.outfile + ".log").c_str());
  std::ostringstream hdr;
  hdr << "#command:   "; for (int i = 0; i < argc; ++i) hdr << argv[i] << " ";
  hdr << "\n#bamfile:   " << o.bamfile << "\n#rdfile:    " << o.rdfile << "\n#reffile:   " << o.reffile << "\n#chrom:     " << o.chr
      << "\n#min_mapq:  " << o.minq << "\n#min_baseQ: " << o.min_baseQ << "\n#binsize:   " << o.P.m << "\n#adjustGC:  " << o.P.gcadjust
      << "\n#output:    " << o.outfile << "\n";
  std::cerr << hdr.str(); log << hdr.str();

  // chromosomes to process: -c RNAME, or (BAM input, rsi.cpp:2114-2131) every reference of the header that is not
  // a mitochondrial / decoy name and has reads
  std::vector<std::string> todo;
  std::vector<int64_t> todo_len;
  if (from_bam && (o.chr.empty() || o.chr == "1-22XY")) {
    std::vector<char> names(1 << 20);
    std::vector<int64_t> lens(1 << 16);
    const int nref = rsi_bam_references(o.bamfile.c_str(), names.data(), (int)names.size(), lens.data(), (int)lens.size());
    if (nref < 0) { std::cerr << rsi_hot_last_error(nullptr) << std::endl; return 0; }
    std::istringstream iss(names.data());
    std::string nm;
    std::cerr << "#Check bam header for 1-22XY \n"; log << "#Check bam header for 1-22XY \n";
    int k = 0;
    while (std::getline(iss, nm)) {
      const int64_t len = k < (int)lens.size() ? lens[(size_t)k] : 0;
      ++k;
      if (nm.find("MT") != std::string::npos || nm.find(".") != std::string::npos) continue;
      todo.push_back(nm); todo_len.push_back(len);
    }
  } else {
    todo.push_back(o.chr); todo_len.push_back(0);
  }
  const bool many = todo.size() > 1;

  // write_cnv_to_file, rsi.cpp:1592-1616: the first processed chromosome opens the file and writes the header, the others
  // append.  Rows always leave in the order of `todo` (the BAM header's), whatever ran where.
  bool wrote_header = false;
  auto emit = [&](const std::string& chr, const ChromOutput& co) {
    std::cerr << co.log; log << co.log;
    if (!co.populated) return;
    std::ofstream out(o.outfile.c_str(), wrote_header ? std::ios::app : std::ios::trunc);
    if (!wrote_header) {
      if (!o.rdfile.empty()) out << "#input " << o.rdfile << " " << chr << std::endl;
      if (from_bam) out << "#input " << o.bamfile << std::endl;
      if (o.P.gcadjust) out << "#GC adjusted\n";
      out << kHeader << std::endl;
      wrote_header = true;
    }
    for (const std::string& r : co.rows) out << r << std::endl;
    out.close();
    std::cerr << "output written to " << o.outfile << std::endl; log << "output written to " << o.outfile << std::endl;
  };

  if (o.gpus <= 0 || !many) {   // the reference's own shape: one chromosome after the other on one context
    int st = 0;
    rsi_ctx* ctx = rsi_hot_create(o.device, &st);
    if (!ctx) { std::cerr << "rsicnv: " << rsi_hot_last_error(nullptr) << std::endl; return 1; }
    for (const std::string& chr : todo) {
      ChromOutput co;
      process_chromosome(ctx, o, chr, many, co);
      emit(chr, co);
      if (co.fatal) break;
    }
    rsi_hot_destroy(ctx);
    return 0;
  }

  // ---- -gpus N: the iterations of the loop are independent.  Chromosomes go to devices longest first
  // (each to the least loaded one), every device gets a pool whose workers take that device's chromosomes longest first;
  // nothing is exchanged between devices but the finished rows, which the main thread writes in header order. ----
  const int ndev = std::max(1, o.gpus), nwork = std::max(1, std::min(o.workers, 32));
  // RSI_HOT_DEVICE_MAP=a,b,...: the HIP device behind logical device 0, 1, ... (default: -gpu, -gpu + 1, ...).  "0,0" runs the
  // two-device code path on a box with one GPU: the partition, the per-device pools and the ordered writer are what is tested.
  std::vector<int> devmap;
  if (const char* dm = getenv("RSI_HOT_DEVICE_MAP")) {
    std::stringstream ss(dm);
    std::string tok;
    while (std::getline(ss, tok, ',')) if (!tok.empty()) devmap.push_back(atoi(tok.c_str()));
  }
  auto physical = [&](int d) { return d < (int)devmap.size() ? devmap[(size_t)d] : o.device + d; };
  std::vector<rsi_pool*> pools;
  for (int d = 0; d < ndev; ++d) {
    int st = 0;
    rsi_pool* pl = rsi_pool_create(physical(d), nwork, &st);
    if (!pl) {
      if (d == 0) { std::cerr << "rsicnv: " << rsi_hot_last_error(nullptr) << std::endl; return 1; }
      std::cerr << "rsicnv: device " << physical(d) << " not available, using " << d << " device(s)" << std::endl;
      break;
    }
    pools.push_back(pl);
  }
  std::cerr << "rsicnv: " << todo.size() << " chromosomes over " << pools.size() << " device(s), " << nwork << " in flight each" << std::endl;
  std::vector<std::vector<int>> per_dev(pools.size());
  {
    std::vector<int> order(todo.size());
    for (size_t i = 0; i < todo.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return todo_len[(size_t)a] > todo_len[(size_t)b]; });
    std::vector<int64_t> load(pools.size(), 0);
    for (int i : order) {
      size_t best = 0;
      for (size_t d = 1; d < pools.size(); ++d) if (load[d] < load[best]) best = d;
      per_dev[best].push_back(i);
      load[best] += todo_len[(size_t)i];
    }
  }
  std::vector<ChromOutput> outs(todo.size());
  std::vector<std::thread> threads;
  std::vector<std::atomic<int>> next(pools.size());
  for (auto& a : next) a = 0;
  for (size_t d = 0; d < pools.size(); ++d)
    for (int w = 0; w < nwork; ++w)
      threads.emplace_back([&, d, w]() {
        rsi_ctx* ctx = rsi_pool_worker(pools[d], w);
        for (;;) {
          const int k = next[d].fetch_add(1);
          if (k >= (int)per_dev[d].size()) break;
          const int i = per_dev[d][(size_t)k];
          process_chromosome(ctx, o, todo[(size_t)i], true, outs[(size_t)i]);
        }
      });
  for (auto& t : threads) t.join();
  for (size_t i = 0; i < todo.size(); ++i) emit(todo[i], outs[i]);
  for (rsi_pool* pl : pools) rsi_pool_destroy(pl);
  return 0;
}
