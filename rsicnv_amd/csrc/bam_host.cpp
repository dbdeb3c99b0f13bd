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

// ---- read-pair annotation (cnv_stat, pairrd.cpp:622-748) ----
#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace rsih {

bool BamReader::seek(uint64_t voff, std::string& err) {
  buf_.clear(); cur_ = 0; eof_ = false;
  next_coff_ = voff >> 16;
  const size_t skip = (size_t)(voff & 0xffff);
  if (!fill(skip, err)) return err.empty() && skip == 0;
  cur_ = skip;
  return true;
}

bool BamReader::fill(size_t need, std::string& err) {
  while (buf_.size() - cur_ < need) {
    if (eof_) return false;
    BgzfBlock b;
    err.clear();
    if (!f_.block_at(next_coff_, b, err)) { eof_ = true; return false; }
    if (cur_ > (1u << 20)) { buf_.erase(buf_.begin(), buf_.begin() + (long)cur_); cur_ = 0; }
    const size_t at = buf_.size();
    buf_.resize(at + b.isize);
    if (!f_.inflate(b, buf_.data() + at, err)) return false;
    next_coff_ += b.csize;
  }
  return true;
}

int BamReader::next(BamRecord& r, std::string& err) {
  if (!fill(4, err)) return err.empty() ? 0 : -1;
  const uint32_t bs = le32(buf_.data() + cur_);
  if (bs < 32) { err = "malformed BAM record"; return -1; }
  if (!fill(4 + (size_t)bs, err)) { if (err.empty()) err = "truncated BAM record"; return -1; }
  const uint8_t* b = buf_.data() + cur_ + 4;
  r.tid = (int32_t)le32(b); r.pos = (int32_t)le32(b + 4);
  const int l_name = b[8];
  r.mapq = b[9];
  r.n_cigar = (int)le16(b + 12); r.flag = (int)le16(b + 14);
  r.l_seq = (int32_t)le32(b + 16);
  r.mtid = (int32_t)le32(b + 20); r.mpos = (int32_t)le32(b + 24); r.isize = (int32_t)le32(b + 28);
  // the variable-length fields must lie inside the record (a corrupt file must not send the CIGAR walk past the buffer)
  if (32ull + (uint64_t)l_name + 4ull * (uint64_t)r.n_cigar > (uint64_t)bs) { err = "malformed BAM record (name / CIGAR overrun the record)"; return -1; }
  int64_t end = r.pos;
  const uint8_t* cig = b + 32 + l_name;
  for (int k = 0; k < r.n_cigar; ++k) {
    const uint32_t c = le32(cig + 4 * k);
    const int op = (int)(c & 0xf);
    if (op == 0 || op == 2 || op == 3) end += c >> 4;     // M, D, N (bam.c:23)
  }
  r.calend = end;
  cur_ += 4 + (size_t)bs;
  return 1;
}

bool bai_linear_offset(const std::string& bai_path, int tid, int64_t pos, uint64_t& voff) {
  FILE* f = fopen(bai_path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  auto rd = [&](void* dst, size_t n) { return fread(dst, 1, n, f) == n; };
  do {
    uint8_t h[8];
    if (!rd(h, 8) || memcmp(h, "BAI\1", 4) != 0) break;
    const int n_ref = (int)le32(h + 4);
    if (tid < 0 || tid >= n_ref) break;
    bool bad = false;
    for (int r = 0; r <= tid && !bad; ++r) {
      uint8_t w[4];
      if (!rd(w, 4)) { bad = true; break; }
      const int n_bin = (int)le32(w);
      for (int b = 0; b < n_bin && !bad; ++b) {
        uint8_t bh[8];
        if (!rd(bh, 8)) { bad = true; break; }
        const int n_chunk = (int)le32(bh + 4);
        if (fseek(f, (long)n_chunk * 16, SEEK_CUR) != 0) bad = true;
      }
      if (bad || !rd(w, 4)) { bad = true; break; }
      const int n_intv = (int)le32(w);
      if (r < tid) { if (fseek(f, (long)n_intv * 8, SEEK_CUR) != 0) bad = true; continue; }
      std::vector<uint8_t> io((size_t)n_intv * 8);
      if (n_intv > 0 && !rd(io.data(), io.size())) { bad = true; break; }
      int64_t w0 = pos >> 14;
      if (w0 >= n_intv) w0 = n_intv - 1;
      for (int64_t k = w0; k >= 0; --k) {
        const uint64_t v = le64(io.data() + 8 * (size_t)k);
        if (v != 0) { voff = v; ok = true; break; }
      }
    }
  } while (false);
  fclose(f);
  return ok;
}

namespace {
// the reads bam_iter_query(tid, beg, end) + bam_iter_read yield (bam_index.c:564-706): same chromosome, pos < end,
// overlapping [beg, end), in file order; visit(r) returns false to stop
template <class Visit>
bool for_reads_in(const BamFile& f, const std::string& bai_path, int tid, int64_t beg, int64_t end, Visit visit, std::string& err) {
  if (beg < 0) beg = 0;
  if (end < beg) return true;
  uint64_t voff = 0;
  if (!bai_linear_offset(bai_path, tid, beg, voff) && !bai_first_offset(bai_path, tid, voff)) return true;   // nothing indexed for tid
  BamReader rd(f);
  if (!rd.seek(voff, err)) return err.empty();
  BamRecord r;
  for (;;) {
    const int rc = rd.next(r, err);
    if (rc < 0) return false;
    if (rc == 0) return true;
    if (r.tid != tid || r.pos >= end) return true;
    const int64_t rend = r.n_cigar ? r.calend : (int64_t)r.pos + 1;
    if (!(rend > beg && r.pos < end)) continue;
    if (!visit(r)) return true;
  }
}
}  // namespace

bool bam_pair_sample(const BamFile& f, const std::string& bai_path, int tid, int64_t tid_len, int64_t beg, int64_t end,
                     PairSample& out, std::string& err) {
  // the part of bam_rd_pr_stats that cnv_stat uses: which reads are sampled (pairrd.cpp:136-175) and their insert sizes
  const int sample_len = 1000000;
  size_t count = 0;
  int pos_start = 0, pos_end = -10000;
  double isize = 0.0, isize2 = 0.0, isize_c = 0;
  const bool ok = for_reads_in(f, bai_path, tid, beg, end, [&](const BamRecord& b) {
    if (b.tid < 0) return true;
    if (b.mtid != b.tid && b.mtid > 0) return true;
    if (b.flag & 0x100) return true;
    if (b.flag & 0x400) return true;
    if ((b.flag & 0x2) && b.mtid == b.tid) {
      isize += abs(b.isize);
      isize2 += (double)(int32_t)((uint32_t)b.isize * (uint32_t)b.isize);   // int * int as the reference computes it
      isize_c += 1;
    }
    if (b.pos >= tid_len) return false;
    if (b.calend >= tid_len) return false;
    if (b.pos > pos_end + 1000) { count = 0; pos_start = b.pos; pos_end = b.pos; }
    pos_end = b.pos;
    count++;
    if (count > 1000000 || (pos_end - pos_start) > sample_len) return false;
    return true;
  }, err);
  if (!ok) return false;
  if (isize_c > 2) {
    isize /= isize_c;
    const double sd = sqrt((isize2 - isize_c * isize * isize) / isize_c);
    out.isize = (int)isize;        // bamstat_st keeps ints (pairrd.cpp:52-53)
    out.isize_sd = (int)sd;
  }
  return true;
}

bool bam_annotate_calls(const BamFile& f, const std::string& bai_path, int tid, const PairSample& ps, std::vector<CallSpan>& calls,
                        std::string& err) {
  int DIS = 1000;
  const int ISIZE_mean = ps.isize, ISIZE_std = ps.isize_sd;
  for (CallSpan& c : calls) {
    int beg = c.start, end = c.end;
    const int TYPE = c.type;
    if (beg > end) std::swap(beg, end);
    const int LEN = end - beg + 1;
    DIS = std::max(DIS, LEN);
    DIS = std::min(DIS, 5000);
    int p1e = beg - DIS;
    const int p2e = end + DIS;
    if (p1e < 1) p1e = 1;
    double q0_all = 0, q0_q0 = 0;
    size_t rp = 0;
    const double ratio = 0.5;
    const bool ok = for_reads_in(f, bai_path, tid, p1e, p2e, [&](const BamRecord& b) {
      if (b.n_cigar <= 1) return true;
      if (b.tid < 0) return true;
      const int rbeg = b.pos;
      const int rend = b.n_cigar ? (int)b.calend : b.pos + 1;
      if (rend > beg && rbeg < end) { q0_all += 1; if (b.mapq == 0) q0_q0 += 1; }
      if (b.mtid != b.tid && b.mtid > 0) return true;
      const bool rev = (b.flag & 0x10) != 0, mrev = (b.flag & 0x20) != 0;
      if (!rev && !mrev) return true;      // both forward
      if (rev && mrev) return true;        // both reverse
      int r1 = rend, r2 = b.mpos;
      if (TYPE == 0) {
        if (r2 - r1 < ISIZE_mean + ISIZE_std * 3) return true;
        const int overlap = std::min(r2, end) - std::max(r1, beg);
        if (overlap < 0) return true;
        if (abs(r1 - beg) + abs(r2 - end) < ISIZE_mean + ISIZE_std * 3) { ++rp; return true; }
        if (overlap < LEN * ratio) return true;
        if (overlap < (r2 - r1) * ratio) return true;
        ++rp;
        return true;
      }
      if (TYPE == 1) {
        if (r2 - r1 > ISIZE_mean - ISIZE_std * 3) return true;
        if (abs(r1 - beg) + abs(r2 - end) < ISIZE_mean + ISIZE_std * 3) { ++rp; return true; }
        if (r1 > r2) std::swap(r1, r2);
        const int overlap = std::min(r2, end) - std::max(r1, beg);
        if (overlap < LEN * ratio) return true;
        if (overlap < (r2 - r1) * ratio) return true;
        ++rp;
        return true;
      }
      return true;
    }, err);
    if (!ok) return false;
    c.q0 = q0_q0 / (q0_all + 0.00001);
    c.rp = (int)rp;
  }
  return true;
}

}  // namespace rsih
