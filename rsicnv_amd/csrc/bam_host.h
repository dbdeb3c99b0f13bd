// bam_host.h -- the host half of the BAM path: BGZF blocks, the BAM header, the
// .bai index (only to find where a chromosome's reads start).  No samtools / htslib: the 0.1.18-era
// BAM format is a few fixed little-endian structures.  The per-read work is on the device
// (kernels_io.hip); this side only inflates and finds record boundaries.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace rsih {

struct BgzfBlock { uint64_t coff; uint32_t csize, isize, hdr; };   // file offset, whole block size, inflated size, header length

class BamFile {
 public:
  ~BamFile();
  bool open(const std::string& path, std::string& err);
  uint64_t size() const { return size_; }
  // block starting at file offset `off`; false at the end of the file or on a malformed block (err set)
  bool block_at(uint64_t off, BgzfBlock& b, std::string& err) const;
  // raw-deflate payload of a block into out[b.isize]
  bool inflate(const BgzfBlock& b, uint8_t* out, std::string& err) const;
  // header: reference names / lengths; *first = virtual offset (coff << 16 | in-block offset) of the first record
  bool read_header(std::vector<std::pair<std::string, int64_t>>& refs, uint64_t& first, std::string& err);
 private:
  int fd_ = -1;
  const uint8_t* map_ = nullptr;
  uint64_t size_ = 0;
};

// Smallest chunk start among the bins of reference `tid` in a .bai file (the place its first read sits).
// false when the index is missing / unreadable / has nothing for tid: the caller then scans from the first record.
bool bai_first_offset(const std::string& bai_path, int tid, uint64_t& voff);

}  // namespace rsih

// ---- read-pair annotation of the calls (cnv_stat, pairrd.cpp:622-748): host only ----
namespace rsih {

struct BamRecord {     // the fields the annotation looks at (BAM spec 4.2), decoded from one record
  int32_t tid, pos, mtid, mpos, isize, l_seq;
  int mapq, flag, n_cigar;
  int64_t calend;      // bam_calend (bam.c:17-27): pos + lengths of M / D / N
};

// Sequential record reader with random access by virtual offset (coff << 16 | offset in the inflated block).
class BamReader {
 public:
  explicit BamReader(const BamFile& f) : f_(f) {}
  bool seek(uint64_t voff, std::string& err);
  // 1: record decoded, 0: end of file, -1: error
  int next(BamRecord& r, std::string& err);
 private:
  bool fill(size_t need, std::string& err);   // make `need` bytes available from cur_
  const BamFile& f_;
  std::vector<uint8_t> buf_;   // inflated bytes not yet consumed (front = cur_)
  size_t cur_ = 0;
  uint64_t next_coff_ = 0;
  bool eof_ = false;
};

// Smallest file position of a read overlapping the 16 kb window that contains `pos` (the .bai linear index), walking
// back to the nearest filled window; false without a usable index.
bool bai_linear_offset(const std::string& bai_path, int tid, int64_t pos, uint64_t& voff);

struct PairSample { int isize = -1, isize_sd = -1; };   // bamstat_st defaults (pairrd.cpp:54-61)
// Insert-size statistics as bam_rd_pr_stats computes them (pairrd.cpp:112-238) over reads from `beg` on.
bool bam_pair_sample(const BamFile& f, const std::string& bai_path, int tid, int64_t tid_len, int64_t beg, int64_t end,
                     PairSample& out, std::string& err);

struct CallSpan { int start, end, type; int rp; double q0; };   // in: start/end/type (0 DEL, 1 DUP); out: rp, q0
// cnv_stat for the calls of one chromosome, in list order (its DIS window carries over from call to call).
bool bam_annotate_calls(const BamFile& f, const std::string& bai_path, int tid, const PairSample& ps, std::vector<CallSpan>& calls,
                        std::string& err);

}  // namespace rsih
