// ingest.hip -- the two data formats in front of the path: the depth of one chromosome from a
// text file (load_data_from_text, loaddata.cpp:496-517) or from a BAM (the pileup loop, samfunctions.cpp), built in HBM.
// The file bytes go through pinned double buffers; parsing, per-read work and the difference-array scan are kernels
// (kernels_io.hip); BGZF inflation and the record walk stay on host threads (bam_host.cpp).
#include "pipeline_internal.h"

using namespace rsik;
using namespace rsip;
using rsih::Candidate;
using rsih::Region;

namespace {

// The sequential parse loop (the reference's rules in the reference's order), used when the device
// cannot prove that positions are strictly increasing.
void parse_depth_text_host(const char* p, size_t sz, int64_t size, std::vector<int32_t>& rd, rsi_text_stats* st) {
  const char* end = p + sz;
  auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
  auto parse_int = [&](const char*& q, const char* e, long long& v) {
    while (q < e && blank(*q)) ++q;
    bool neg = false;
    if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; ++q; }
    if (q >= e || *q < '0' || *q > '9') { v = 0; return false; }
    long long x = 0;
    while (q < e && *q >= '0' && *q <= '9') { x = x * 10 + (*q - '0'); ++q; }
    v = neg ? -x : x;
    return true;
  };
  const char* q = p;
  while (q < end) {
    const char* eol = (const char*)memchr(q, '\n', (size_t)(end - q));
    if (!eol) eol = end;
    if (eol > q && *q != '#') {
      const char* c = q;
      long long pos = 0, d = 0;
      if (parse_int(c, eol, pos)) {
        parse_int(c, eol, d);
        if (pos >= 1) {
          ++st->lines;
          if (pos >= size) { ++st->beyond; break; }       // loaddata.cpp:514
          rd[(size_t)pos - 1] = (int32_t)d;
          ++st->stored;
        }
      }
    }
    q = eol + 1;
  }
}

constexpr size_t kTextChunk = size_t(64) << 20;   // bytes of text per transfer + kernel

}  // namespace

extern "C" {

int rsi_hot_load_depth_text(rsi_ctx* ctx, const char* path, int64_t n, rsi_text_stats* stats) {
  rsi_text_stats local;
  rsi_text_stats* st = stats ? stats : &local;
  memset(st, 0, sizeof(*st));
  if (!ctx || !path) return fail(ctx, RSI_ERR_BAD_ARG, "null argument");
  if (n <= 0 || n >= (1ll << 31) - 4096) return fail(ctx, RSI_ERR_BAD_ARG, "chromosome length must be in (0, 2^31)");
  const double t0 = now_ms();
  HIPCHK(hipSetDevice(ctx->device));
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return fail(ctx, RSI_ERR_BAD_ARG, std::string("Cannot open file ") + path);
  struct FdGuard { int fd; ~FdGuard() { close(fd); } } guard{fd};
  struct stat sb;
  if (fstat(fd, &sb) != 0) return fail(ctx, RSI_ERR_BAD_ARG, std::string("Cannot stat file ") + path);
  st->bytes = (int64_t)sb.st_size;
  if (!ctx_enter(ctx)) return RSI_ERR_HIP;
  mailbox_reset(ctx);
  HIPCHK(ctx->in_depth.ensure((size_t)(n + 4) * 4));
  HIPCHK(hipMemsetAsync(ctx->in_depth.p, 0, (size_t)n * 4, ctx->stream));
  ctx->n_in = n;
  if (ctx->text_pin_cap < kTextChunk) {
    for (int b = 0; b < 2; ++b) {
      if (ctx->text_pin[b]) (void)hipHostFree(ctx->text_pin[b]);
      ctx->text_pin[b] = nullptr;
      if (hipHostMalloc(reinterpret_cast<void**>(&ctx->text_pin[b]), kTextChunk, hipHostMallocDefault) != hipSuccess)
        return fail(ctx, RSI_ERR_INTERNAL, "out of pinned host memory for the text staging");
    }
    ctx->text_pin_cap = kTextChunk;
  }
  const int max_wg = text_parse_workgroups((long long)kTextChunk);
  HIPCHK(ctx->text_dev[0].ensure(kTextChunk));
  HIPCHK(ctx->text_dev[1].ensure(kTextChunk));
  HIPCHK(ctx->text_wg.ensure((size_t)max_wg * 16 * 2 + 256));   // (first, max) per workgroup, two chunks in flight, + stats
  uint8_t* wgbase = ctx->text_wg.as<uint8_t>();
  TextParseStats* d_stats = reinterpret_cast<TextParseStats*>(wgbase + (size_t)max_wg * 32);
  HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(TextParseStats), ctx->stream));
  std::vector<long long> wg_host[2];
  wg_host[0].resize((size_t)max_wg * 2); wg_host[1].resize((size_t)max_wg * 2);

  // Double buffering: while the device parses chunk k the host reads chunk k+1 from the file.  A chunk ends on
  // a line end; the partial last line is carried to the front of the next chunk.
  size_t carry = 0;            // bytes of an unfinished line already at the front of the buffer being filled
  bool eof = false, unsorted = false;
  long long run_max = -1;      // largest position seen in the chunks checked so far
  int inflight_wgs[2] = {0, 0};
  hipEvent_t done[2] = {nullptr, nullptr};
  for (int b = 0; b < 2; ++b) if (hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess) return fail(ctx, RSI_ERR_HIP, "hipEventCreate failed");
  struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int b = 0; b < 2; ++b) if (e[b]) (void)hipEventDestroy(e[b]); } } evguard{done};
  auto check_chunk = [&](int b) {   // cross-workgroup order of a finished chunk
    const long long* f = wg_host[b].data();
    const long long* m = f + inflight_wgs[b];
    for (int w = 0; w < inflight_wgs[b]; ++w) {
      if (f[w] < 0) continue;
      if (run_max >= 0 && f[w] <= run_max) unsorted = true;
      run_max = m[w] > run_max ? m[w] : run_max;
    }
    inflight_wgs[b] = 0;
  };
  int cur = 0;
  bool used[2] = {false, false};
  while (!eof) {
    char* buf = ctx->text_pin[cur];   // free: its previous chunk was waited for before `carry` was parked in it
    size_t have = carry;
    while (have < kTextChunk) {
      const ssize_t got = read(fd, buf + have, kTextChunk - have);
      if (got < 0) return fail(ctx, RSI_ERR_INTERNAL, std::string("read error on ") + path);
      if (got == 0) { eof = true; break; }
      have += (size_t)got;
    }
    size_t len = have;
    if (!eof) {   // cut at the last line end
      while (len > 0 && buf[len - 1] != '\n') --len;
      if (len == 0) return fail(ctx, RSI_ERR_UNSUPPORTED, "a line of the depth file is longer than 64 MB");
    }
    if (len > 0) {
      const int nwg = text_parse_workgroups((long long)len);
      long long* d_first = reinterpret_cast<long long*>(wgbase + (size_t)cur * max_wg * 16);
      long long* d_max = d_first + nwg;
      HIPCHK(hipMemcpyAsync(ctx->text_dev[cur].p, buf, len, hipMemcpyHostToDevice, ctx->stream));
      { Timer t(ctx, "parse_depth_text"); launch_parse_depth_text(ctx->text_dev[cur].p, (long long)len, (long long)n, ctx->in_depth.as<int32_t>(), d_first, d_max, d_stats, ctx->stream); }
      HIPCHK(hipMemcpyAsync(wg_host[cur].data(), d_first, (size_t)nwg * 16, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipEventRecord(done[cur], ctx->stream));
      inflight_wgs[cur] = nwg;
      used[cur] = true;
    }
    // the other buffer's chunk (the older one) has to be through before the unfinished line is parked in it
    // and the next read fills it; the chunk just launched keeps the device busy meanwhile
    const int other = cur ^ 1;
    if (used[other]) { HIPCHK(hipEventSynchronize(done[other])); check_chunk(other); used[other] = false; }
    carry = have - len;
    if (carry) memcpy(ctx->text_pin[other], buf + len, carry);
    cur = other;
  }
  for (int b = 0; b < 2; ++b) if (used[b] && inflight_wgs[b]) { HIPCHK(hipEventSynchronize(done[b])); check_chunk(b); }
  TextParseStats hs;
  HIPCHK(hipMemcpyAsync(&hs, d_stats, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  st->lines = (int64_t)hs.lines; st->stored = (int64_t)hs.stored; st->beyond = (int64_t)hs.beyond;
  if (hs.unsorted || unsorted) {
    // order-dependent rules in play: redo the file with the sequential loop
    st->fallback = 1; st->lines = st->stored = st->beyond = 0;
    std::vector<int32_t> rd((size_t)n, 0);
    std::vector<char> all((size_t)st->bytes);
    if (lseek(fd, 0, SEEK_SET) != 0) return fail(ctx, RSI_ERR_INTERNAL, "seek error");
    size_t have = 0;
    while (have < all.size()) { const ssize_t got = read(fd, all.data() + have, all.size() - have); if (got <= 0) break; have += (size_t)got; }
    parse_depth_text_host(all.data(), have, n, rd, st);
    HIPCHK(hipMemcpyAsync(ctx->in_depth.p, rd.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  st->t_total_ms = now_ms() - t0;
  if (ctx->timing) { double tot = 0; for (const KernelTime& k : ctx->ktimes) { float ms = 0; (void)hipEventElapsedTime(&ms, k.a, k.b); tot += ms; } st->t_parse_kernel_ms = tot; }
  return RSI_OK;
}

int rsi_hot_run_text(rsi_ctx* ctx, const rsi_params* p, const char* depth_path, const uint8_t* fasta, int64_t n, rsi_result** out,
                     rsi_text_stats* stats) {
  if (!ctx || !p || !depth_path || !fasta || !out) return fail(ctx, RSI_ERR_BAD_ARG, "null argument");
  ctx->ktimes.clear(); ctx->event_next = 0;
  int rc = rsi_hot_load_depth_text(ctx, depth_path, n, stats);
  if (rc != RSI_OK) return rc;
  HIPCHK(ctx->in_fasta.ensure((size_t)n + 64));
  HIPCHK(hipMemcpyAsync(ctx->in_fasta.p, fasta, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return rsi_hot_run_device(ctx, p, ctx->in_depth.p, ctx->in_fasta.p, n, out);
}


int rsi_hot_load_depth_bam(rsi_ctx* ctx, const char* bam_path, const char* chrom, int minq, int min_baseq, rsi_bam_stats* stats) {
  rsi_bam_stats local;
  rsi_bam_stats* st = stats ? stats : &local;
  memset(st, 0, sizeof(*st));
  if (!ctx || !bam_path || !chrom) return fail(ctx, RSI_ERR_BAD_ARG, "null argument");
  const double t0 = now_ms();
  HIPCHK(hipSetDevice(ctx->device));
  std::string err;
  rsih::BamFile bam;
  if (!bam.open(bam_path, err)) return fail(ctx, RSI_ERR_BAD_ARG, err);
  std::vector<std::pair<std::string, int64_t>> refs;
  uint64_t voff = 0;
  if (!bam.read_header(refs, voff, err)) return fail(ctx, RSI_ERR_BAD_ARG, err);
  int tid = -1;
  for (size_t r = 0; r < refs.size(); ++r) if (refs[r].first == chrom) tid = (int)r;
  if (tid < 0) return fail(ctx, RSI_ERR_BAD_ARG, std::string("chromosome not in the BAM header: ") + chrom);
  const int64_t n = refs[(size_t)tid].second;
  if (n <= 0 || n >= (1ll << 31) - 4096) return fail(ctx, RSI_ERR_BAD_ARG, "chromosome length must be in (0, 2^31)");
  st->tid = tid; st->n = n;
  uint64_t idx_off = 0;
  if (rsih::bai_first_offset(std::string(bam_path) + ".bai", tid, idx_off)) { voff = idx_off; st->indexed = 1; }

  if (!ctx_enter(ctx)) return RSI_ERR_HIP;

  mailbox_reset(ctx);
  HIPCHK(ctx->in_depth.ensure((size_t)(n + 4) * 4));
  int32_t* d_diff = ctx->in_depth.as<int32_t>();     // difference array first, scanned in place into the depth
  HIPCHK(hipMemsetAsync(d_diff, 0, (size_t)(n + 1) * 4, ctx->stream));
  ctx->n_in = n;
  if (ctx->text_pin_cap < kTextChunk) {
    for (int b = 0; b < 2; ++b) {
      if (ctx->text_pin[b]) (void)hipHostFree(ctx->text_pin[b]);
      ctx->text_pin[b] = nullptr;
      if (hipHostMalloc(reinterpret_cast<void**>(&ctx->text_pin[b]), kTextChunk, hipHostMallocDefault) != hipSuccess)
        return fail(ctx, RSI_ERR_INTERNAL, "out of pinned host memory for the BAM staging");
    }
    ctx->text_pin_cap = kTextChunk;
  }
  HIPCHK(ctx->text_dev[0].ensure(kTextChunk));
  HIPCHK(ctx->text_dev[1].ensure(kTextChunk));
  constexpr size_t kMaxRec = kTextChunk / 36 + 16;   // a record is at least 36 bytes
  const size_t stats_off = 2 * kMaxRec * 4, scan_off = stats_off + 256;
  HIPCHK(ctx->text_wg.ensure(scan_off + (size_t)scan_tiles(n) * 4 + 64));
  uint8_t* wsb = ctx->text_wg.as<uint8_t>();
  BamDepthStats* d_stats = reinterpret_cast<BamDepthStats*>(wsb + stats_off);
  HIPCHK(hipMemsetAsync(d_stats, 0, sizeof(BamDepthStats), ctx->stream));

  hipEvent_t done[2] = {nullptr, nullptr};
  for (int b = 0; b < 2; ++b) if (hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess) return fail(ctx, RSI_ERR_HIP, "hipEventCreate failed");
  struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int b = 0; b < 2; ++b) if (e[b]) (void)hipEventDestroy(e[b]); } } evguard{done};
  bool used[2] = {false, false};
  std::vector<uint32_t> rec_off[2];
  rec_off[0].reserve(kMaxRec / 8); rec_off[1].reserve(kMaxRec / 8);
  const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));

  // Pipeline over chunks of inflated bytes: while the host walks the records of chunk k and hands them to the device,
  // the inflate threads already fill the other buffer with chunk k+1.  The inflated data of a chunk starts at kReserve,
  // so that the unfinished record at the end of chunk k can be parked right in front of chunk k+1 without waiting.
  constexpr size_t kReserve = size_t(8) << 20;
  struct Chunk {
    std::vector<rsih::BgzfBlock> blocks;
    std::vector<size_t> at;
    // speculative record walk of each block, done by the thread that inflated it, as if the block began on a record
    // boundary (htslib and samtools flush before a record that would not fit, so normally it does): offsets of the
    // records of `tid` that lie wholly inside the block, how many records were seen, where the walk stopped, and
    // whether it met a read beyond `tid` (the end of the chromosome in a sorted file)
    struct BlockWalk { std::vector<uint32_t> offs; uint32_t seen = 0; size_t stop = 0; bool beyond = false, bad = false; };
    std::vector<BlockWalk> walks;
    size_t end = kReserve;     // end of the inflated data in the buffer
    bool eof = false, failed = false;
    std::string err;
    double t_inflate = 0;
    std::thread worker;
  } chunk[2];
  uint64_t coff = voff >> 16;            // next block to inflate
  auto prepare = [&](int b) -> bool {    // choose the blocks of the next chunk and start inflating them into buffer b
    Chunk& c = chunk[b];
    c.blocks.clear(); c.at.clear(); c.end = kReserve; c.eof = false; c.failed = false; c.err.clear();
    for (;;) {
      rsih::BgzfBlock blk;
      std::string e2;
      if (!bam.block_at(coff, blk, e2)) { if (!e2.empty()) { c.failed = true; c.err = e2; return false; } c.eof = true; break; }
      if (c.end + blk.isize > kTextChunk) break;
      c.blocks.push_back(blk); c.at.push_back(c.end);
      c.end += blk.isize; coff += blk.csize;
      st->bytes_compressed += blk.csize;
    }
    if (c.blocks.empty() && !c.eof) { c.failed = true; c.err = "a BGZF block does not fit the staging buffer"; return false; }
    uint8_t* buf = reinterpret_cast<uint8_t*>(ctx->text_pin[b]);
    c.walks.assign(c.blocks.size(), Chunk::BlockWalk());
    c.worker = std::thread([&c, buf, &bam, nthreads, tid]() {
      const double ti = now_ms();
      std::atomic<size_t> next(0);
      std::mutex emu;
      auto work = [&]() {
        for (;;) {
          const size_t k = next.fetch_add(1);
          if (k >= c.blocks.size()) break;
          std::string e2;
          if (!bam.inflate(c.blocks[k], buf + c.at[k], e2)) { std::lock_guard<std::mutex> lk(emu); c.failed = true; c.err = e2; continue; }
          Chunk::BlockWalk& w = c.walks[k];
          size_t p = c.at[k];
          const size_t lim = c.at[k] + c.blocks[k].isize;
          while (p + 8 <= lim) {
            const uint32_t bs = (uint32_t)buf[p] | ((uint32_t)buf[p + 1] << 8) | ((uint32_t)buf[p + 2] << 16) | ((uint32_t)buf[p + 3] << 24);
            if (bs < 32) { w.bad = true; break; }          // not a record start after all (or a broken file): the checker decides
            if (p + 4 + (size_t)bs > lim) break;
            const int32_t rtid = (int32_t)((uint32_t)buf[p + 4] | ((uint32_t)buf[p + 5] << 8) | ((uint32_t)buf[p + 6] << 16) | ((uint32_t)buf[p + 7] << 24));
            ++w.seen;
            if (rtid == tid) w.offs.push_back((uint32_t)(p - c.at[k]));
            else if (rtid > tid || rtid < 0) { w.beyond = true; break; }
            p += 4 + (size_t)bs;
          }
          w.stop = p;
        }
      };
      std::vector<std::thread> th;
      const unsigned nt = (unsigned)std::min<size_t>(nthreads, c.blocks.size());
      for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
      work();
      for (auto& t : th) t.join();
      c.t_inflate = now_ms() - ti;
    });
    return true;
  };
  struct JoinGuard { Chunk* c; ~JoinGuard() { for (int b = 0; b < 2; ++b) if (c[b].worker.joinable()) c[b].worker.join(); } } joinguard{chunk};

  size_t skip = (size_t)(voff & 0xffff); // bytes of the first block that precede the first record
  size_t carry = 0;                      // bytes of an unfinished record parked in front of the current chunk's data
  bool finished = false;
  int cur = 0;
  if (!prepare(0)) return fail(ctx, RSI_ERR_BAD_ARG, chunk[0].err);
  while (!finished) {
    Chunk& c = chunk[cur];
    c.worker.join();
    if (c.failed) return fail(ctx, RSI_ERR_BAD_ARG, c.err);
    st->t_inflate_ms += c.t_inflate;
    st->bytes_inflated += (int64_t)(c.end - kReserve);
    const int other = cur ^ 1;
    // the other buffer is free once the device has taken its previous chunk: start the next inflate right away
    { const double tq = now_ms(); if (used[other]) { HIPCHK(hipEventSynchronize(done[other])); used[other] = false; } st->t_wait_ms += now_ms() - tq; }
    bool more = false;
    if (!c.eof) { if (!prepare(other)) return fail(ctx, RSI_ERR_BAD_ARG, chunk[other].err); more = true; }
    // ---- record boundaries; the BAM is coordinate sorted, so reading ends with the first read beyond `tid` ----
    uint8_t* buf = reinterpret_cast<uint8_t*>(ctx->text_pin[cur]);
    const double tw = now_ms();
    std::vector<uint32_t>& offs = rec_off[cur];
    offs.clear();
    const size_t start = kReserve - carry + skip;   // `skip` only applies to the very first chunk (no carry there)
    skip = 0;
    size_t p = start;
    const size_t have = c.end;
    size_t kb = 0;                       // first block that starts at or after p
    while (p + 4 <= have && !finished) {
      while (kb < c.blocks.size() && c.at[kb] < p) ++kb;
      if (kb < c.blocks.size() && c.at[kb] == p && !c.walks[kb].bad) {
        // the block does begin on a record boundary: its thread has walked it already
        const Chunk::BlockWalk& w = c.walks[kb];
        const uint32_t base = (uint32_t)(p - start);
        for (uint32_t o : w.offs) offs.push_back(base + o);
        st->records += w.seen; st->on_chrom += (int64_t)w.offs.size();
        if (w.beyond) { finished = true; p = w.stop; break; }
        if (w.stop == p) {               // not even one whole record in this block: walk it the plain way below
        } else { p = w.stop; ++kb; continue; }
      }
      const uint32_t bs = (uint32_t)buf[p] | ((uint32_t)buf[p + 1] << 8) | ((uint32_t)buf[p + 2] << 16) | ((uint32_t)buf[p + 3] << 24);
      if (bs < 32) return fail(ctx, RSI_ERR_BAD_ARG, "malformed BAM record");
      if (p + 4 + bs > have) break;
      const int32_t rtid = (int32_t)((uint32_t)buf[p + 4] | ((uint32_t)buf[p + 5] << 8) | ((uint32_t)buf[p + 6] << 16) | ((uint32_t)buf[p + 7] << 24));
      ++st->records;
      if (rtid == tid) { offs.push_back((uint32_t)(p - start)); ++st->on_chrom; }
      else if (rtid > tid || rtid < 0) { finished = true; break; }
      p += 4 + (size_t)bs;
    }
    if (c.eof) finished = true;
    st->t_walk_ms += now_ms() - tw;
    const size_t len = p;                // end of the whole records examined
    if (!offs.empty()) {
      uint32_t* d_off = reinterpret_cast<uint32_t*>(wsb + (size_t)cur * kMaxRec * 4);
      HIPCHK(hipMemcpyAsync(ctx->text_dev[cur].p, buf + start, len - start, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(d_off, offs.data(), offs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
      { Timer t(ctx, "bam_depth"); launch_bam_depth(ctx->text_dev[cur].p, d_off, (int)offs.size(), tid, minq, min_baseq, (long long)n, d_diff, d_stats, ctx->stream); }
      HIPCHK(hipEventRecord(done[cur], ctx->stream));
      used[cur] = true;
    }
    carry = finished ? 0 : have - len;
    if (carry > kReserve) return fail(ctx, RSI_ERR_UNSUPPORTED, "a BAM record is longer than 8 MB");
    if (carry && more) memcpy(ctx->text_pin[other] + (kReserve - carry), buf + len, carry);   // in front of the data being inflated there
    if (!more) finished = true;
    cur = other;
  }
  { Timer t(ctx, "depth_scan"); launch_inclusive_scan_i32(d_diff, (long long)n, reinterpret_cast<int32_t*>(wsb + scan_off), ctx->stream); }
  BamDepthStats hs;
  HIPCHK(hipMemcpyAsync(&hs, d_stats, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
  { const double tq = now_ms(); HIPCHK(hipStreamSynchronize(ctx->stream)); st->t_wait_ms += now_ms() - tq; }
  st->used = (int64_t)hs.used; st->runs = (int64_t)hs.runs; st->malformed = (int64_t)hs.malformed;
  st->t_total_ms = now_ms() - t0;
  return RSI_OK;
}

int rsi_bam_references(const char* bam_path, char* names, int names_cap, int64_t* lengths, int max_refs) {
  if (!bam_path) return RSI_ERR_BAD_ARG;
  std::string err;
  rsih::BamFile bam;
  std::vector<std::pair<std::string, int64_t>> refs;
  uint64_t voff = 0;
  if (!bam.open(bam_path, err) || !bam.read_header(refs, voff, err)) { set_global_error(err); return RSI_ERR_BAD_ARG; }
  std::string all;
  for (size_t r = 0; r < refs.size(); ++r) {
    if (r) all += '\n';
    all += refs[r].first;
    if (lengths && (int)r < max_refs) lengths[r] = refs[r].second;
  }
  if (names && names_cap > 0) { strncpy(names, all.c_str(), (size_t)names_cap - 1); names[names_cap - 1] = 0; }
  return (int)refs.size();
}

int rsi_hot_run_bam(rsi_ctx* ctx, const rsi_params* p, const char* bam_path, const char* chrom, int minq, int min_baseq,
                    const uint8_t* fasta, int64_t n, rsi_result** out, rsi_bam_stats* stats) {
  if (!ctx || !p || !bam_path || !chrom || !fasta || !out) return fail(ctx, RSI_ERR_BAD_ARG, "null argument");
  rsi_bam_stats local;
  rsi_bam_stats* st = stats ? stats : &local;
  ctx->ktimes.clear(); ctx->event_next = 0;
  int rc = rsi_hot_load_depth_bam(ctx, bam_path, chrom, minq, min_baseq, st);
  if (rc != RSI_OK) return rc;
  if (st->n != n) return fail(ctx, RSI_ERR_BAD_ARG, "reference and target not same size (loaddata.cpp:284-287)");
  HIPCHK(ctx->in_fasta.ensure((size_t)n + 64));
  HIPCHK(hipMemcpyAsync(ctx->in_fasta.p, fasta, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return rsi_hot_run_device(ctx, p, ctx->in_depth.p, ctx->in_fasta.p, n, out);
}

}  // extern "C"
