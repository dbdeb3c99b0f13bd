// kernels.h -- launch wrappers of the gfx950 kernels behind librsi_hot.so.
// Every wrapper enqueues on `stream` and returns without synchronising.  Pointers are device
// pointers unless the name says otherwise.  Kernel list and the roofline that bounds each one:
// DESIGN.md section 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "device_util.h"

namespace rsik {

// A rejected launch (too much dynamic LDS, a bad grid) raises no error where it happens: the wrappers below return nothing.
// Every launch of the library therefore goes through RSI_LAUNCH, which reads the runtime's last error right behind the launch
// and keeps the first failure in a thread-local slot; the pipeline's next wait (ctx_sync) collects it.  A context's launches all
// come from the thread that runs it, and the entry points drain whatever the host application's own calls left in the runtime's
// slot before the first launch (drain_stale_errors), so only this library's launches are attributed to a run.
inline void note_launch() {   // tl_launch_error: device_util.h
  const hipError_t e = hipGetLastError();   // cleared by the read
  if (e != hipSuccess && tl_launch_error == hipSuccess) tl_launch_error = e;
}
inline void drain_stale_errors() { (void)hipGetLastError(); tl_launch_error = hipSuccess; }
inline hipError_t take_launch_error() { const hipError_t e = tl_launch_error; tl_launch_error = hipSuccess; return e; }
#define RSI_LAUNCH(...) do { hipLaunchKernelGGL(__VA_ARGS__); rsik::note_launch(); } while (0)

constexpr int kGcLevels = 202;        // window GC count 0..201 (gccontent.cpp:102, 115)
constexpr int kHistValues = 65536;    // directly indexed integer histogram range
constexpr int kResClasses = 32;       // 31 MAD residues (rsi.cpp:1130) + 1 class for the tail
constexpr int kTileBases = 4096;      // bases per workgroup tile in the streaming kernels

// ---- K1: FASTA bytes -> GC bitmask + N bitmask (loaddata.cpp:481-483; readref.cpp:95) ----
// gcbits/nbits hold nwords = n/64 + 1 words, bit j of word w <-> base 64*w + j, zero beyond n.
// `fill`: ranges cleared for the kernels that follow in the stream (K1 is the first kernel of a chromosome's chain)
void launch_fasta_classify(const uint8_t* fasta, int64_t n, uint64_t* gcbits, uint64_t* nbits, int64_t nwords,
                           const FillList& fill, hipStream_t stream);
void launch_fill(const FillList& fill, hipStream_t stream);   // the same clearing as a launch of its own (rare paths)
// Scratch for the in-kernel folds of the per-workgroup slabs (device_util.h, fold_slabs): group sums of the widest slab.
// counters: kFoldGroups + 1 arrival counters per kernel, zero before the launch (every launch leaves them zero).
size_t fold_scratch_bytes();
// What the per-base kernels hand each other on the device, so that K4j can be queued behind K2j without the host in between:
// the regions K4j removes (K1b's last workgroup), the cap (K2j's last workgroup), and K4j's verdict on its launch configuration.
struct PhaseParams {
  int32_t nreg;        // padded, merged N regions (cbreak / cum in device memory)
  int32_t capval;      // (int)(cap median * cap), or < 0: not known on the device (escapes pending, deep coverage, wrapped counters)
  int64_t ncompact;    // n minus the removed bases
  int32_t regions_ok;  // 0: the boundary list was too long or malformed for the device: the host's regions count
  int32_t redo;        // set by K4j: launched with a configuration that does not fit the cap -- nothing written, launch again
  int32_t pad[2];
};
// K1b's arguments, for the launch that does its work on the way (K2j, round 5): the N mask, the boundary list and its count (zero before),
// the padding of get_noseq_regions (loaddata.cpp:243-273), where the compacted break points and removed lengths go
struct NRuns { const uint64_t* nbits; uint64_t* list; uint32_t* count; uint32_t cap; int32_t dx; int64_t* cbreak; int64_t* cum; };
// ---- K1b: run boundaries of the N bitmask -> unordered list of (pos << 1 | is_end) ----
// pp != NULL: the last workgroup also builds cbreak[4100] / cum[4097] (regions padded by dx, merged) and fills pp's region fields;
// counter: an arrival counter, zero before and after.
void launch_n_transitions(const uint64_t* nbits, int64_t nwords, uint64_t* list, uint32_t* count, uint32_t cap, int64_t n, int dx,
                          PhaseParams* pp, int64_t* cbreak, int64_t* cum, unsigned int* counter, hipStream_t stream);

// ---- K2: GC table accumulation (checkgccontent pass 1, gccontent.cpp:105-145) ----
struct GcAccum {
  unsigned long long sum[kGcLevels];   // sum of depth per window GC count
  unsigned long long cnt[kGcLevels];
  unsigned long long possum, poscnt;   // over depth > 0
  unsigned int negatives;              // bit 0: depth < 0 seen (unsupported); bit 1: depth >= 2^21, packed form invalid
  unsigned int escapes;                // bases of kByteEscape and more (saturating count): their byte copy is the escape code
};
// K2 leaves a byte copy of the depth behind (depth8[i] = min(depth[i], kByteEscape)): the later per-base passes stream one
// byte per base instead of four and turn to the int32 array only where a byte says kByteEscape.
constexpr int kByteEscape = 255;
// K3' leaves a byte copy of the GC-RESCALED depth behind, saturated at kByteSat ("this much or more"): K4' streams it when
// the cap is below that (a saturated value is capped either way).
constexpr int kByteSat = 254;
// packed = 1: one LDS atomic per base (count and sum in one 64-bit word), valid for depths < 2^21;
// when the result carries flag bit 1 the caller zeroes acc and launches again with packed = 0.
// slabs: scratch of gc_hist_slab_bytes(n) bytes (per-workgroup partial results, folded by a second tiny kernel).
// The last workgroup to finish folds the slabs into acc, adds the last n % 4 bases and builds table[kGcLevels + 1]
// (level means, then the mean of the positive depths): gccontent.cpp:109-112, 141-145.
size_t gc_hist_slab_bytes(int64_t n);
void launch_gc_hist(const int32_t* depth, const uint64_t* gcbits, int64_t n, GcAccum* acc, double* table, int packed, void* slabs,
                    void* gsum, unsigned int* counters, uint8_t* depth8 /* n + 16 bytes, or NULL */, hipStream_t stream);

// ---- K3: GC rescale + value histogram (adjustgccontent, gccontent.cpp:43-92; feeds apply_cap) ----
// table[202] and rdmean as computed on the host from GcAccum.  out may be NULL (histogram only);
// adjust=0 copies depth through unchanged (the -NOGC path only needs the histogram).
// hist[kHistValues] counts output values < kHistValues; *big counts the rest; *vmax: the largest value of 256 and more seen
// (0: none) -- counters beyond it are zero, which bounds the median walk.
struct ValueHistAux { unsigned long long big; unsigned int vmax; unsigned int negatives; };
// Median walk of partition_stat_tp (wufunctions.cpp:398-420, dy = 1) over hist[kHistValues] for `total` values, on the device:
// inrange = sum of the counters, lo / hi = smallest / largest value present (lo > hi: none), med = the bucket where the
// cumulated count first reaches total/2 (-1: never).
struct ValueMedian { unsigned long long inrange; int32_t lo, hi, med, pad; };
// ---- K2j: K2 with the joint histogram [window GC count][depth byte] (kernels_base.hip): its last workgroup builds the GC table,
// the value histogram of the RESCALED depth (what K3' computed per base), walks it to the cap median (*vm) and hands the
// header over (head_src -> head_dst, mapped host memory) -- the chromosome needs no K3'.  acc->negatives bit 2: a
// workgroup's 16-bit counters wrapped, the launch's results are void (the caller runs K2 + K3' instead); acc->escapes > 0:
// the histogram lacks the depths of 255 and more until launch_escape_hist has run.  totals: gc_joint_totals_bytes(), zero
// before the launch; hist / aux zero before the launch; counters: kFoldGroups + 1 arrival counters.
// esc_list: gc_joint_esc_list_bytes() of scratch (the workgroups' first escapes by position: the last workgroup adds them to the
// histogram itself; info->esc_pending says when there were too many for that).  rtab: kGcLevels words, per level the 16.16
// fixed-point ratio K4j rescales depth bytes with (verified against the reference's expression for every byte; bit 31: not
// usable, take the exact expression).
struct JointInfo { int32_t gmin, gmax;   // GC levels that occur
                   int32_t vmax;         // largest depth byte below kByteEscape that occurs
                   int32_t esc_pending;  // 1: the value histogram lacks the escapes (launch_escape_hist), or the coverage is deep (int32 kernels)
};
size_t gc_joint_slab_bytes(int64_t n);
size_t gc_joint_totals_bytes();
size_t gc_joint_esc_list_bytes();
void launch_gc_joint_hist(const int32_t* depth, const uint64_t* gcbits, int64_t n, GcAccum* acc, double* table, void* slabs, void* totals,
                          unsigned int* counters, uint8_t* depth8 /* n + 2048 bytes */, uint32_t* hist, ValueHistAux* aux, ValueMedian* vm,
                          const void* head_src, void* head_dst, size_t head_bytes, void* esc_list, unsigned int* rtab, JointInfo* info,
                          PhaseParams* pp /* capval (and redo = 0) for a K4j queued behind */, double cap_mult, hipStream_t stream,
                          const NRuns* nruns = nullptr /* K1b's work inside this launch (see NRuns) */);
void launch_escape_hist(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table, uint32_t* hist,
                        ValueHistAux* aux, unsigned int* counter, ValueMedian* vm, const void* head_src, void* head_dst, size_t head_bytes,
                        hipStream_t stream);

size_t gc_rescale_slab_bytes(int64_t n);   // scratch for the per-workgroup histograms
// table: kGcLevels level means followed by the mean of the positive depths (K2's last workgroup builds it).
// The last workgroup to finish folds the slabs into hist, fixes the tail quirks of the 20-slice write-back (SURVEY App. A
// Q2/Q3) in out[] and hist[], adds the last n % 4 bases (which the streaming loop leaves out), walks hist to *vm and
// copies head_bytes from head_src (device) to head_dst (mapped host memory; NULL: no copy).
void launch_gc_rescale(const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                       int adjust, int32_t* out, uint32_t* hist, ValueHistAux* aux, void* slabs, void* gsum,
                       unsigned int* counters, ValueMedian* vm, const void* head_src, void* head_dst, size_t head_bytes,
                       hipStream_t stream, uint8_t* out8 = nullptr,
                       PhaseParams* pp = nullptr /* the launch's last workgroup leaves the cap there (median x cap_mult, or -1) for a K4 queued behind it */,
                       double cap_mult = 0.0);
// Only the rescaled array: out[] as the launch above leaves it, nothing else touched (rsi_hot_fetch of "rd_gc" when the run
// itself streamed the byte copy and never wrote the int32 array).
void launch_gc_materialize(const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table, int32_t* out,
                           unsigned int* counter, hipStream_t stream);
// The value histogram of the rescaled depth from the byte copy, nothing written per base (what apply_cap's median needs,
// loaddata.cpp:233); same last-workgroup work as launch_gc_rescale.
size_t value_hist8_slab_bytes(int64_t n);
void launch_value_hist8(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                        uint32_t* hist, ValueHistAux* aux, void* slabs, void* gsum, unsigned int* counters, ValueMedian* vm,
                        const void* head_src, void* head_dst, size_t head_bytes, uint8_t* rescaled8 /* n + 2048 bytes */,
                        const unsigned int* escapes /* K2's count of bases that did not fit a byte (GcAccum::escapes) */, hipStream_t stream);
// Above this many escaped bases (n / 8) the byte path is not worth taking: K3' only hands the header over and the host
// runs the int32 kernels (deep coverage, 250x and more).
unsigned int byte_escape_limit(int64_t n);

// ---- K4: cap + N-region compaction + per-bin median/sum + chromosome statistics ----
// (apply_cap loaddata.cpp:229; concatenate_data loaddata.cpp:48; _median/variance rsi.cpp:2202;
//  median_transfer rsi.cpp:1363; bin sums + MAD subsamples rsi.cpp:1127-1157)
struct BinAccum {
  unsigned long long big;        // values >= kHistValues (not in the histograms): unsupported by the caller
  unsigned int vmax;
  unsigned int pad;
};
// cbreak[k]: compacted index where region k is cut out; cum[k]: bases removed before compacted
// index cbreak[k] (cum[nreg] = total).  res_hist: [kHistValues][kResClasses] counts of value by
// class (compacted index mod 31, or 31 for the tail beyond 31*floor(n'/31)); the chromosome's sum,
// sum of squares and median all derive from it.
// slabs: scratch of cap_compact_slab_bytes(...) bytes for the per-workgroup histograms.
// The last workgroup to finish folds the slabs into res_hist and copies exp_bytes from exp_src (device) to exp_dst
// (mapped host memory).  cap_compact_overwrites(): every value is below the LDS range (the cap is), res_hist is
// overwritten and needs no clearing; otherwise it must be zero before the launch.  Region lists of up to kRegInline
// entries travel in `inl` with the kernel arguments (cbreak / cum may then be NULL).
constexpr int kRegInline = 48;
struct K4Regions { long long brk[kRegInline]; long long cum[kRegInline + 1]; };
size_t cap_compact_slab_bytes(int m, int32_t capval, int64_t ncompact, int vbase);
int cap_compact_overwrites(int m, int32_t capval, int64_t ncompact, int vbase);
// First value of the window the LDS histograms of the int32 kernels cover: 0 unless the distribution's centre (mean / median
// depth) is 160 and more, then centre - 3/8 width; values outside the window are counted with global atomics, slowly but
// exactly.  K3 derives its window on the device (1024 values); K4's (kK4Window = 512 values) comes from the host.
constexpr int kK4Window = 512;
int hist_window_base(double center, int width);
void launch_cap_compact_bin(const int32_t* src, int64_t n, const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg,
                            int64_t ncompact, int32_t capval, int m, int32_t* rdc, int32_t* binmed, int64_t* binsum,
                            uint32_t* res_hist, BinAccum* acc, void* slabs, void* gsum, unsigned int* counters,
                            const void* exp_src, void* exp_dst, size_t exp_bytes, int vbase /* hist_window_base(cap median), or 0 */,
                            hipStream_t stream);

// K4w: the same for caps of 254 .. 32766 (deep coverage): int32 in, 16-bit tile and 16-bit window counters in LDS, int32 out.
// res_hist must be zero before the launch (the window's sums and the stray values are added to it); vbase as for K4.
int cap_compact16_applies(int m, int32_t capval, int64_t ncompact);
size_t cap_compact16_slab_bytes(int m, int64_t ncompact);
void launch_cap_compact_bin16(const int32_t* src, const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg, int64_t ncompact,
                              int32_t capval, int m, int32_t* rdc, int32_t* binmed, int64_t* binsum, uint32_t* res_hist, void* slabs,
                              unsigned int* counters, const void* exp_src, void* exp_dst, size_t exp_bytes, int vbase, hipStream_t stream);

// K4 fed from K3''s byte copy of the rescaled depth (rescaled8): the rescaled int32 array is never needed.  Applies when
// cap_compact8_applies(): 1 <= capval < kByteSat (every capped value fits a byte, res_hist is overwritten) and m <= 104.
// depth / gcbits / table (K2's [kGcLevels] level means + the mean of the positive depths): for the tiles at the chromosome's
// ends and across removed regions, which recompute the rescale per element.
int cap_compact8_applies(int m, int32_t capval);
size_t cap_compact8_slab_bytes(int m, int32_t capval, int64_t ncompact);
void launch_cap_compact_bin8(const uint8_t* rescaled8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                             const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg, int64_t ncompact, int32_t capval,
                             int m, uint8_t* rdc8 /* ncompact + 64 bytes: the capped, compacted depth as bytes */, int32_t* binmed, int64_t* binsum,
                             uint32_t* res_hist, void* slabs, void* gsum,
                             unsigned int* counters, const void* exp_src, void* exp_dst, size_t exp_bytes, hipStream_t stream, int raw = 0);

// K4j: K4' fed from the byte copy of the RAW depth (K2 / K2j's depth8), rescaling on the way with the GC table -- same outputs.
void launch_rescale_compact_bin8(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                                 const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg, int64_t ncompact, int32_t capval,
                                 int m, uint8_t* rdc8, int32_t* binmed, int64_t* binsum, uint32_t* res_hist, void* slabs, void* gsum,
                                 unsigned int* counters, const void* exp_src, void* exp_dst, size_t exp_bytes,
                                 const unsigned int* rtab /* K2j's fixed-point ratios, or NULL: the float form with its exactness margin */,
                                 PhaseParams* pp /* NULL, or: nreg / ncompact / capval are read from it on the device (cbreak / cum from device
                                                    memory); the arguments of those names only shape the launch (capval: a guess, checked) */,
                                 hipStream_t stream);

// K4 as two launches (kernels_k4s.hip): K4s -- rescale, cap, compaction, byte store, residue-class histogram, wave-autonomous at
// eight waves per SIMD, fixed point in the loop, the chunks it cannot do that way (region cuts, chromosome ends, escape bytes, GC
// levels without a verified ratio) exactly behind it -- and K4m -- the bins' medians and sums from the bytes K4s leaves, no LDS.
// Together the arguments and results of launch_rescale_compact_bin8 with rtab (no gsum: the groups' sums are atomic adds) wherever
// rescale_compact_split_applies(); slabs: rescale_compact_split_slab_bytes(); rdc8: rescale_compact_split_rdc_bytes() (padded to
// whole sub-tiles); escapes: K2j's count of depths of 255 and more (GcAccum::escapes, device memory).
int rescale_compact_split_applies(int m, int32_t capval, int64_t ncompact, int nreg);
size_t rescale_compact_split_slab_bytes(int32_t capval, int64_t ncompact);
size_t rescale_compact_split_rdc_bytes(int64_t ncompact);
void launch_rescale_compact_stream(const uint8_t* depth8, const int32_t* depth, const uint64_t* gcbits, int64_t n, const double* table,
                                   const int64_t* cbreak, const int64_t* cum, const K4Regions& inl, int nreg, int64_t ncompact, int32_t capval,
                                   int m, uint8_t* rdc8, uint32_t* res_hist, void* slabs, unsigned int* counters, const void* exp_src, void* exp_dst,
                                   size_t exp_bytes, const unsigned int* rtab, const unsigned int* escapes, PhaseParams* pp, hipStream_t stream);
void launch_bin_median8(const uint8_t* rdc8, int64_t ncompact, int32_t capval, int m, int32_t* binmed, int64_t* binsum, const PhaseParams* pp,
                        hipStream_t stream);

// ---- K5: NB variance-stabilising transform (negative_binomial_transfer, rsi.cpp:1155-1185) ----
// raw[b] = (float)(2 sqrt(r) log(sqrt(q) + sqrt(1+q))), q = (sum+0.25)/(m2*r-0.5); *rawmin_bits =
// complement of the smallest order key over the bins (atomicMax; zero on entry = nothing seen).
void launch_nb_raw(const int64_t* binsum, int64_t nb, int m, int64_t ncompact, double r, float* raw,
                   uint32_t* rawmin_bits, hipStream_t stream);
// ---- K6: 0.01-grid histogram quantiles of float arrays (partition_stat_tp, wufunctions.cpp:364) ----
// min_inv = complement of the smallest order key seen, max_bits = largest key; all zero = nothing seen yet
struct MinMaxF { uint32_t min_inv, max_bits; unsigned int nonfinite; unsigned int pad; };
// The whole median as a chain of two launches (min/max + grid plan by the last workgroup -> histogram + walk by the last
// workgroup); `out` receives the result, or the flag of the first test that failed.
struct GridMedian { double med, ymin; unsigned long long count; uint32_t np, flags; };
enum { kGridEmpty = 1, kGridNonFinite = 2, kGridDegenerate = 4, kGridTooWide = 8 };
// Shared state of a context's chains: mm holds "nothing seen" (all zero) before a chain and again after it; hist has cap
// entries; counters: two arrival counters (plan, walk), zero before and after.
struct GridChain { MinMaxF* mm; uint32_t* hist; uint32_t cap; unsigned int* counters; };
// what the last workgroup of launch_hist_walk copies to mapped host memory (NULL dst: nothing)
struct GridExport { const void* src[2]; void* dst[2]; size_t bytes[2]; };

// x = (float)(raw - tmin); x = (float)(x / med_nbt * med); bins 0..2 <- the three reference levels (rsi.cpp:1176-1185), where
// tmin is the minimum launch_nb_raw left in *rawmin_bits and med_raw / del_raw / dup_raw are the transform of the median,
// half and one-and-a-half times the median depth computed by the host (the kernel and the host derive the scaled levels
// with the same IEEE operations).  The values' min/max are taken on the way and the last workgroup plans the median's grid
// into *out: first link of a chain, continue with launch_hist_walk.
void launch_nb_scale_minmax(float* x, int64_t nb, const uint32_t* rawmin_bits, double med_raw, double del_raw, double dup_raw,
                            double RDmedian, const GridChain& c, GridMedian* out, hipStream_t stream);
// -MED: out_f = (float)in, min/max of |out_f - center| taken on the way (first link of the MAD chain)
void launch_i32_to_f32_minmax(const int32_t* in, float* out_f, int64_t nb, double center, const GridChain& c, GridMedian* out,
                              hipStream_t stream);
// min/max over x[i] (or |x[i]-center| rounded to float when use_abs), restricted to mask[i]==0 when mask != NULL
void launch_minmax_f32(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, MinMaxF* mm,
                       hipStream_t stream);
// hist[(size_t)((v - ymin)/0.01 + 0.5)] += 1, same selection as above; hist has np entries
void launch_hist_f32(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, double ymin,
                     uint32_t* hist, uint32_t np, hipStream_t stream);
// first link: min/max of the selection, grid planned into *out.  d_center != NULL: the centre is read from device memory
// (an earlier chain's `med`).
void launch_minmax_plan(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, const double* d_center,
                        const GridChain& c, GridMedian* out, hipStream_t stream);
// second link: histogram on the planned grid, median into *out; then the export, and `fill` for the kernels behind it
void launch_hist_walk(const float* x, const int32_t* mask, int64_t nb, int use_abs, double center, const double* d_center,
                      const GridChain& c, GridMedian* out, const GridExport* ex, const FillList* fill, hipStream_t stream);

// ---- K7: RSI scan (rsistatus, rsi.cpp:1191-1259; runmeantp wufunctions.cpp:573-647) ----
struct ScanParams {
  int64_t nb;
  int32_t Lmax;
  int16_t pad;
  int16_t kcap;      // set by launch_rsi_scan: highest block level of the mark tables
  double tmedian;
  double lim_del;    // 0.75 * RDmedian
  double lim_dup;    // 1.25 * RDmedian
};
// Longest scan: the reference's Lmax is max(20, 10000/m, cal_max) (rsi.cpp:1830-1831): 10000 at -m 1.  Up to kScanLdsL the exact
// sweep's tile (256 bins + a halo of Lmax/2 + 1 each side, staged with its rank and mark tables) fits the 160 KB of LDS; longer
// scans keep the same code on a tile in device memory (a workspace slice per workgroup): slow per tile, but the detection pass
// in front sends only the tiles that can hit there.  Staged indices are 16 bits.
constexpr int kMaxScanL = 10400;
constexpr int kScanLdsL = 3800;
constexpr int kScanPad = 8;   // thr_del / thr_dup carry this many unreachable entries (-inf / +inf) after index Lmax
// thr_del[L], thr_dup[L] (L = 1..Lmax, index L): a window of length L is a DEL hit iff
// sum <= thr_del[L], a DUP hit iff sum >= thr_dup[L] (host-derived, see scan_thresholds()).
// first_del / first_dup: smallest L that marks the bin, 0xffffffff when none.  counters[0] = trim
// escapes, counters[1] = values breaking the exact-sum precondition.
constexpr int kThrInline = 232;   // thresholds that fit the kernel arguments: Lmax + 1 + kScanPad <= kThrInline
struct ScanThr { double del[kThrInline]; double dup[kThrInline]; };
// inl != NULL: thresholds in the kernel arguments (thr_del / thr_dup unused)
// Two launches (kernels_bin.hip): a detection pass lists the tiles of 256 bins in which any lane can hit at any L (float
// prefixes against thresholds widened by the rounding bound: a superset), the exact sweep then runs on the listed tiles only,
// each tile's lengths split over several workgroups.  tiles: scratch for (nb / 256 + 1) tile indices; NULL = no detection
// pass, every tile through the exact sweep.  counters: the pass' work block (zero before): [0] trim escapes, [1] values
// breaking the exact-sum precondition, [8] tiles listed, [9] the exact sweep's task counter.
void launch_rsi_scan(const float* T, const int32_t* medint, const ScanParams& sp, const double* thr_del,
                     const double* thr_dup, const ScanThr* inl, uint32_t* first_del, uint32_t* first_dup, uint32_t* counters,
                     uint32_t* tiles, void* tile_ws /* scan_tile_workspace_bytes(Lmax) when that is not 0 */, hipStream_t stream);
size_t scan_tile_workspace_bytes(int Lmax);   // 0: the exact sweep's tile fits LDS
// Stop levels of the two sweeps (rsi.cpp:1225, 1255; DEL marks win, App. A Q14) in one launch.
// work: [16 uint32: escapes, inexact (the scan's), ldel, ldup, both-count, ...][hist_del][hist_dup] (scan_level_stride(Lmax) words each), zero
// before the scan; both: scratch for kBothCap (first_del, first_dup) pairs.  The last workgroup copies host_bytes of work to
// host_copy (mapped host memory).
constexpr int kMaxLevels = 10416;   // >= kMaxScanL + 1 + kScanPad, a multiple of 4
inline int scan_level_stride(int Lmax) { return (Lmax + 1 + kScanPad + 3) & ~3; }   // hist_dup sits this many words behind hist_del
constexpr size_t kScanWorkBytes = 64 + 2 * (size_t)kMaxLevels * 4;
void launch_level_stop(const uint32_t* first_del, const uint32_t* first_dup, int64_t nb, int32_t Lmax, uint32_t* work, void* both,
                       unsigned int* counter, void* host_copy, size_t host_bytes, hipStream_t stream);
// status[j] = -first_del[j] if first_del[j] <= levels[0]; else +first_dup[j] if <= levels[1]; else 0; copy (may be NULL)
// receives the same.  Run boundaries appended unordered to runs as (pos << 1 | is_end), *count entries (zero before);
// host_copy (may be NULL): [count, 0][first host_entries entries] written by the last workgroup.
void launch_resolve_runs(const uint32_t* first_del, const uint32_t* first_dup, const uint32_t* levels, int64_t nb, int32_t* status,
                         int32_t* copy, uint64_t* runs, uint32_t* count, uint32_t cap, unsigned int* counter, void* host_copy,
                         uint32_t host_entries, hipStream_t stream);

// ---- K9/K10: marked runs and max-score sub-segment (get_continuous_segments rsi.cpp:291;
//      get_rsi_segments rsi.cpp:1060) ----
struct BestSeg { double score; int32_t start; int32_t len; };
struct SegItem { int32_t run; int32_t len; int32_t Lbeg; int32_t Lend; };   // lengths [Lbeg, Lend) of one run
// Short run / item lists ride in the kernel arguments (the pointer forms are for longer ones).
constexpr int kRunsInline = 200, kItemsInline = 150;
struct RunsInline { int32_t se[2 * kRunsInline]; int32_t off[kRunsInline]; };     // (start, end) pairs; prefix offsets
struct ItemsInline { SegItem it[kItemsInline]; int32_t off[kRunsInline]; };
// exact double prefix of every run into scratch + poff[run] (len+1 entries), one workgroup per run
// status_out (may be NULL; may be mapped host memory): the runs' status values, run after run, value e of run r at poff[r] - r + e
void launch_run_prefix(const float* T, const int32_t* run_start, const int32_t* run_end, const RunsInline* inl, int nruns,
                       const int64_t* poff, double* scratch, const int32_t* status, int32_t* status_out, hipStream_t stream);
// one workgroup per work item; out[item] = best (score, offset, L) of that item under the
// reference's visiting order (larger score, then smaller L, then smaller offset); out may be mapped host memory
void launch_best_items(const void* items, const ItemsInline* inl, int nitems, const int64_t* poff, const double* scratch,
                       double tmedian, BestSeg* out, hipStream_t stream);
// edge trimming of filterstatus (rsi.cpp:1023-1044): one thread per run
void launch_trim_runs(const float* T, int32_t* status, const int32_t* run_start, const int32_t* run_end, const RunsInline* inl,
                      int nruns, double delthr, double addthr, hipStream_t stream);

// ---- filterstatus' per-level float sums on the device (kernels_fs.hip; rsi.cpp:967-976, App. A Q13) ----
// out: [2 Lmax + 1] float sums then [2 Lmax + 1] int32 counts, index = level + Lmax: the sum of T over the bins of each
// status level accumulated in FLOAT in index order, bit for bit what the sequential loop gives (an exact parallel form, see
// the file).  A count of -1 at level 0: the device declined (negative / non-finite values, or more than clist_cap marked
// bins) and the caller runs the loop itself.  ws: level_sums_workspace_bytes() laid out for scans up to Lmax_cap; its first
// level_sums_head_bytes(Lmax_cap) bytes zero before the launch.  counter: one arrival counter, zero before and after.
// host_copy: mapped host memory for `out` (2 (2 Lmax + 1) words).
size_t level_sums_head_bytes(int Lmax_cap);
size_t level_sums_workspace_bytes(int64_t nb, int32_t clist_cap, int Lmax_cap);
void launch_level_sums(const float* T, const int32_t* status, int64_t nb, int Lmax, void* ws, int Lmax_cap, int32_t clist_cap, float* out,
                       unsigned int* counter, void* host_copy, hipStream_t stream);

// Dynamic LDS beyond 48 KB has to be allowed per kernel and per device.  Done once per (kernel, device) and for all the CU
// has left next to the kernel's static use: a per-launch setting from several host threads (a pool) would race with the
// other threads' launches.  A failed attribute call is reported (stderr) -- the launch that needed it then fails and the
// pipeline's next wait returns the launch error.
constexpr int kMaxDevices = 16;
void report_attribute_failure(const char* kernel, const char* what);
#define RSI_ALLOW_FULL_LDS(kernel)                                                                                  \
  do {                                                                                                              \
    static std::once_flag once__[rsik::kMaxDevices];                                                                \
    int dev__ = 0;                                                                                                  \
    if (hipGetDevice(&dev__) != hipSuccess || dev__ < 0 || dev__ >= rsik::kMaxDevices) dev__ = 0;                   \
    std::call_once(once__[dev__], [dev__] {                                                                         \
      hipFuncAttributes a__;                                                                                        \
      int lds__ = 0;                                                                                                \
      if (hipFuncGetAttributes(&a__, reinterpret_cast<const void*>(kernel)) != hipSuccess) {                        \
        rsik::report_attribute_failure(#kernel, "hipFuncGetAttributes"); return;                                    \
      }                                                                                                             \
      if (hipDeviceGetAttribute(&lds__, hipDeviceAttributeMaxSharedMemoryPerBlock, dev__) != hipSuccess || lds__ <= 0) \
        lds__ = 160 * 1024;                                                                                         \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                              lds__ - (int)a__.sharedSizeBytes) != hipSuccess)                                      \
        rsik::report_attribute_failure(#kernel, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");                 \
    });                                                                                                             \
  } while (0)

// ---- candidate stages on the device (kernels_cand.hip) ----
constexpr unsigned kCandHistBins = 16384;   // LDS counters of the per-test quantile histograms
struct EdgeJob {            // optimize_with_derivative for one candidate; start/end updated in place by each launch
  int32_t start, end, type, pad;
};
struct CandJob {            // one isitcnvwrap test, prepared on the host from the candidate list
  int32_t start, end;       // the candidate (compacted coordinates, inclusive)
  int32_t kind;             // 0 DEL, 1 DUP
  int32_t margin;           // int(len*buffer + 1)
  int32_t capacity;         // (int)(chklen*d*2): slots of the neighbourhood array
  int32_t top;              // slot the left side starts filling from (downwards)
  int32_t nleft, nright;    // neighbour intervals the walk may have to jump over, in pointer order
  int32_t left_off, right_off;   // offsets into the chain array (int2 start,end)
  int32_t budget;           // maxchkbp*10: thinning threshold
  int32_t cut;              // bit 0 / 1: the left / right chain was cut short by the host (the kernel reports when it runs out)
  double right_cap;         // 2*chklen*d (the right side appends while used < right_cap)
  int64_t iscratch_off;     // int32 scratch: (top+1) + capacity + min(capacity, budget), each rounded up to a multiple of 4; offset a multiple of 4
  int64_t lscratch_off;     // int64 scratch: capacity + 1 rounded up to a multiple of 4; offset even
};
struct CandOut {
  int32_t flags;            // 1: empty neighbourhood, 2: body value range beyond the LDS histogram, 4: same for the window means, 8: cut chain used up, 16: the left walk ran short where the host had vouched it would not (split form)
  int32_t nref, nbody, nwin, left_reach, right_reach, body_min, body_max;
  double body_q[3], body_s1, body_s2;   // lower quartile / median / upper quartile (partition_stat_tp), sum, sum of squares
  double ref_q[3], ref_s1, ref_s2;      // the same for the window means
};
// The capped, compacted depth as the candidate kernels see it: int32 (bytes = 4) or, behind K4', one byte per base (bytes = 1).
struct DepthRef { const void* p; int bytes; };
void launch_widen_u8(const uint8_t* src, int64_t n, int32_t* dst, hipStream_t stream);   // the int32 form of a byte array
void launch_patch_i32(int32_t* dst, const int32_t* pos, const int32_t* val, int64_t cnt, hipStream_t stream);   // dst[pos[k]] = val[k]
void launch_range_sums(DepthRef rdc, const void* ranges /* int2 lo,hi inclusive */, int nranges, long long* sums, hipStream_t stream);
// one launch = one call of optimize_with_derivative for every job; ws: sharpen_workspace_bytes(ws_jobs) bytes laid out for
// ws_jobs >= njobs jobs, whose first sharpen_workspace_zero_bytes(ws_jobs) are zero before the first launch (each launch
// leaves them zero again, so a workspace is cleared once, when it is allocated).  jobs may be mapped host memory.
void launch_sharpen_edges(DepthRef rdc, int64_t ncompact, EdgeJob* jobs, int njobs, void* ws, int ws_jobs, hipStream_t stream);
size_t sharpen_workspace_bytes(int njobs);
size_t sharpen_workspace_zero_bytes(int njobs);
// The same test spread over several workgroups (four launches: the two walks side by side; chunked prefix + the candidate's
// statistics; chunked window means; chunked histogram with a last-workgroup fold).  iscratch carries a fourth piece per job
// (the right walk's own `capacity` slots, after the three of the single-workgroup form); mid: njobs CandMid records, `done`
// zero before the first launch (each launch leaves it zero); ghist: njobs x kCandHistBins counters, zero likewise.
constexpr int kCandChunks = 16;
struct CandMid {
  int32_t lcnt, lreach, lused, rcnt_max, rreach, rused;
  uint32_t done, body_flags;
  long long totals[kCandChunks];
  float flo[kCandChunks], fhi[kCandChunks];
  double m1[kCandChunks], m2[kCandChunks];
  int32_t body_min, body_max;
  double body_q[3], body_s1, body_s2;
};
void launch_candidate_test_split(DepthRef rdc, int64_t ncompact, const CandJob* jobs, int njobs, const void* chains,
                                 int32_t* iscratch, long long* lscratch, double RDmedian, CandMid* mid, uint32_t* ghist,
                                 CandOut* outs, hipStream_t stream);
void launch_candidate_test(DepthRef rdc, int64_t ncompact, const CandJob* jobs, int njobs, const void* chains,
                           int32_t* iscratch, long long* lscratch, double RDmedian, CandOut* outs, hipStream_t stream);

// ---- depth text ingestion (kernels_io.hip; load_data_from_text's parse loop, loaddata.cpp:496-517) ----
struct TextParseStats { unsigned long long lines, stored, beyond; unsigned int unsorted, pad; };
// One chunk of text that starts on a line start and ends on a line end (or the end of the file).  depth[size] must be
// zeroed before the first chunk.  wg_first / wg_max: text_parse_workgroups(nbytes) entries each (first counted position
// and running maximum per workgroup; -1 = none) for the caller's cross-workgroup order check; stats accumulates.
int text_parse_workgroups(long long nbytes);
void launch_parse_depth_text(const void* text, long long nbytes, long long size, int32_t* depth, long long* wg_first,
                             long long* wg_max, TextParseStats* stats, hipStream_t stream);

// ---- BAM pileup -> depth (kernels_io.hip; load_data_from_bam, loaddata.cpp:277-333 + resolve_cigar_pos, samfunctions.cpp:38-100) ----
struct BamDepthStats { unsigned long long used, runs, malformed; };   // malformed: records whose fields overrun their block_size (skipped)
// One thread per record: data = inflated BAM bytes, rec_off[i] = offset of record i's block_size field.  diff: int32[n + 1], zeroed
// before the first chunk; launch_inclusive_scan_i32 over n turns it into the depth (tile_scratch: scan_tiles(n) ints).
void launch_bam_depth(const void* data, const uint32_t* rec_off, int nrec, int tid, int minq, int min_baseq, long long n, int32_t* diff,
                      BamDepthStats* stats, hipStream_t stream);
int scan_tiles(long long n);
void launch_inclusive_scan_i32(int32_t* x, long long n, int32_t* tile_scratch, hipStream_t stream);

}  // namespace rsik
