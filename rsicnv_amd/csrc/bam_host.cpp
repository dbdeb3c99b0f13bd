// bam_host.cpp -- see bam_host.h.  Format references: SAM/BAM specification v1 sections 4.1 (BGZF), 4.2 (BAM),
// 5.2 (BAI); the reference reads the same files through its vendored samtools-0.1.18 (bgzf.c, bam_import.c, bam_index.c).
#include "bam_host.h"
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <stdio.h>

namespace rsih {

namespace {
inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t le16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
}  // namespace

BamFile::~BamFile() {
  if (map_) munmap(const_cast<uint8_t*>(map_), size_);
  if (fd_ >= 0) close(fd_);
}

bool BamFile::open(const std::string& path, std::string& err) {
  fd_ = ::open(path.c_str(), O_RDONLY);
  if (fd_ < 0) { err = "Cannot open file " + path; return false; }
  struct stat sb;
  if (fstat(fd_, &sb) != 0 || sb.st_size < 28) { err = "not a BAM file: " + path; return false; }
  size_ = (uint64_t)sb.st_size;
  void* p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
  if (p == MAP_FAILED) { err = "Cannot map file " + path; return false; }
  map_ = static_cast<const uint8_t*>(p);
  (void)madvise(p, size_, MADV_SEQUENTIAL);
  return true;
}

bool BamFile::block_at(uint64_t off, BgzfBlock& b, std::string& err) const {
  if (off + 18 > size_) return false;
  const uint8_t* p = map_ + off;
  if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) { err = "malformed BGZF block header"; return false; }
  const uint32_t xlen = le16(p + 10);
  uint32_t bsize = 0;
  bool found = false;
  for (uint32_t x = 0; x + 4 <= xlen && off + 12 + x + 4 <= size_;) {      // extra subfields: find 'B','C'
    const uint8_t* e = p + 12 + x;
    const uint32_t slen = le16(e + 2);
    if (e[0] == 'B' && e[1] == 'C' && slen == 2) { bsize = le16(e + 4) + 1; found = true; break; }
    x += 4 + slen;
  }
  if (!found || off + bsize > size_ || bsize < 12 + xlen + 8) { err = "malformed BGZF block"; return false; }
  b.coff = off; b.csize = bsize; b.hdr = 12 + xlen; b.isize = le32(p + bsize - 4);
  return true;
}

bool BamFile::inflate(const BgzfBlock& b, uint8_t* out, std::string& err) const {
  if (b.isize == 0) return true;
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2 failed"; return false; }
  zs.next_in = const_cast<Bytef*>(map_ + b.coff + b.hdr);
  zs.avail_in = b.csize - b.hdr - 8;
  zs.next_out = out;
  zs.avail_out = b.isize;
  const int rc = ::inflate(&zs, Z_FINISH);
  inflateEnd(&zs);
  if (rc != Z_STREAM_END || zs.avail_out != 0) { err = "BGZF block does not inflate to its stated size"; return false; }
  return true;
}

bool BamFile::read_header(std::vector<std::pair<std::string, int64_t>>& refs, uint64_t& first, std::string& err) {
  // the header may span blocks: inflate blocks into one growing buffer until it is complete
  std::vector<uint8_t> buf;
  std::vector<uint64_t> block_start_u;   // inflated offset at which each block starts
  std::vector<uint64_t> block_coff;
  uint64_t off = 0;
  auto need = [&](size_t bytes) {
    while (buf.size() < bytes) {
      BgzfBlock b;
      if (!block_at(off, b, err)) { if (err.empty()) err = "truncated BAM header"; return false; }
      block_start_u.push_back(buf.size()); block_coff.push_back(off);
      buf.resize(buf.size() + b.isize);
      if (!inflate(b, buf.data() + buf.size() - b.isize, err)) return false;
      off += b.csize;
    }
    return true;
  };
  if (!need(12)) return false;
  if (memcmp(buf.data(), "BAM\1", 4) != 0) { err = "not a BAM file (magic)"; return false; }
  const uint32_t l_text = le32(buf.data() + 4);
  size_t p = 8 + (size_t)l_text;
  if (!need(p + 4)) return false;
  const uint32_t n_ref = le32(buf.data() + p);
  p += 4;
  refs.clear();
  for (uint32_t r = 0; r < n_ref; ++r) {
    if (!need(p + 4)) return false;
    const uint32_t l_name = le32(buf.data() + p);
    if (!need(p + 4 + l_name + 4)) return false;
    std::string name(reinterpret_cast<const char*>(buf.data() + p + 4), l_name ? l_name - 1 : 0);
    const int64_t len = (int32_t)le32(buf.data() + p + 4 + l_name);
    refs.push_back({name, len});
    p += 4 + l_name + 4;
  }
  // virtual offset of inflated position p
  size_t k = block_start_u.size() - 1;
  while (k > 0 && block_start_u[k] > p) --k;
  if (p == buf.size()) { first = off << 16; }                    // header ends exactly at a block end: records start with the next block
  else first = (block_coff[k] << 16) | (uint64_t)(p - block_start_u[k]);
  return true;
}

bool bai_first_offset(const std::string& bai_path, int tid, uint64_t& voff) {
  FILE* f = fopen(bai_path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  uint8_t h[8];
  auto rd = [&](void* dst, size_t n) { return fread(dst, 1, n, f) == n; };
  do {
    if (!rd(h, 8) || memcmp(h, "BAI\1", 4) != 0) break;
    const int n_ref = (int)le32(h + 4);
    if (tid < 0 || tid >= n_ref) break;
    bool bad = false;
    for (int r = 0; r <= tid && !bad; ++r) {
      uint8_t w[4];
      if (!rd(w, 4)) { bad = true; break; }
      const int n_bin = (int)le32(w);
      uint64_t best = ~0ull;
      for (int b = 0; b < n_bin && !bad; ++b) {
        uint8_t bh[8];
        if (!rd(bh, 8)) { bad = true; break; }
        const uint32_t bin = le32(bh);
        const int n_chunk = (int)le32(bh + 4);
        for (int c = 0; c < n_chunk; ++c) {
          uint8_t ch[16];
          if (!rd(ch, 16)) { bad = true; break; }
          if (bin != 37450 && le64(ch) < best) best = le64(ch);      // 37450: the metadata pseudo-bin of later samtools
        }
      }
      if (bad) break;
      if (!rd(w, 4)) { bad = true; break; }
      const int n_intv = (int)le32(w);
      if (fseek(f, (long)n_intv * 8, SEEK_CUR) != 0) { bad = true; break; }
      if (r == tid && best != ~0ull) { voff = best; ok = true; }
    }
  } while (false);
  fclose(f);
  return ok;
}

}  // namespace rsih
