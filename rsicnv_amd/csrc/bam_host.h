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
