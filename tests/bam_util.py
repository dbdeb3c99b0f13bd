"""Test plumbing for the BAM path: a minimal BAM writer (BGZF + records, enough for the
fields the reference's loader looks at), synthetic read sets, the reference's depth rules in numpy-free
Python for small cases, and helpers around the compiled reference (index + `-s` depth dump)."""
import os
import struct
import subprocess
import zlib

import numpy as np

CIGAR_OPS = "MIDNSHP=X"
FLAG_SECONDARY, FLAG_DUP, FLAG_REVERSE, FLAG_UNMAP = 0x100, 0x400, 0x10, 0x4


def _bgzf_block(data):
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = comp.compress(data) + comp.flush()
    bsize = len(body) + 25
    head = struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord("B"), ord("C"), 2, bsize)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14: return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17: return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20: return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23: return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26: return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def encode_read(tid, pos0, mapq, flag, cigar, seq_len, qual, name=b"r", mtid=-1, mpos=-1, tlen=0):
    """cigar: list of (op_char, length); qual: bytes of length seq_len; pos0: 0-based leftmost position."""
    ref_len = sum(l for op, l in cigar if op in "MDN=X")
    cig = b"".join(struct.pack("<I", (l << 4) | CIGAR_OPS.index(op)) for op, l in cigar)
    seq = bytes([0x11] * ((seq_len + 1) // 2))          # all 'A'
    nm = name + b"\0"
    core = struct.pack("<iiBBHHHiiii", tid, pos0, len(nm), mapq, reg2bin(pos0, pos0 + max(ref_len, 1)), len(cigar), flag, seq_len, mtid, mpos, tlen)
    body = core + nm + cig + seq + bytes(qual)
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, records, block=60000, straddle=False):
    """refs: list of (name, length); records: encoded reads in coordinate order.  straddle: cut the BGZF blocks at
    fixed sizes wherever that falls (records, even their length fields, then span blocks -- legal, and what a reader
    has to survive), instead of flushing before a record that would not fit (what samtools / htslib write)."""
    text = "@HD\tVN:1.0\tSO:coordinate\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(refs))
    for n, l in refs:
        hdr += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    with open(path, "wb") as f:
        f.write(_bgzf_block(hdr))
        buf = b""
        if straddle:
            data = b"".join(records)
            for a in range(0, len(data), block):
                f.write(_bgzf_block(data[a:a + block]))
            records = []
        for r in records:
            if len(buf) + len(r) > block and buf:       # records may also straddle blocks: every 7th flush splits one
                f.write(_bgzf_block(buf)); buf = b""
            buf += r
            if len(buf) > block:
                f.write(_bgzf_block(buf[:block])); buf = buf[block:]
        if buf:
            f.write(_bgzf_block(buf))
        f.write(_bgzf_block(b""))                        # EOF marker


def synth_reads(n, coverage=8.0, read_len=100, seed=1, tid=0, ntid=1):
    """Coordinate-sorted reads over a chromosome of n bases with the CIGAR / flag / quality variety the
    loader's rules react to.  Returns the encoded records (all tids interleaved in file order)."""
    rng = np.random.default_rng(seed)
    out = []
    for t in range(ntid):
        nreads = int(n * coverage / read_len)
        starts = np.sort(rng.integers(0, max(1, n - read_len // 2), nreads))
        starts[:3] = [0, 0, 1]                           # pos == 0 is skipped by the reference (loaddata.cpp:315)
        for i, p in enumerate(starts.tolist()):
            kind = rng.integers(0, 20)
            L = read_len
            if kind == 0:   cig = [("S", 10), ("M", L - 10)]
            elif kind == 1: cig = [("M", 40), ("I", 5), ("M", L - 45)]
            elif kind == 2: cig = [("M", 30), ("D", 7), ("M", L - 30)]
            elif kind == 3: cig = [("M", 25), ("N", 200), ("M", L - 25)]
            elif kind == 4: cig = [("=", 50), ("X", 2), ("=", L - 52)]     # =/X: the reference's position quirk
            elif kind == 5: cig = [("H", 5), ("M", L - 20), ("S", 20)]
            elif kind == 6: cig = []                                       # no CIGAR (unmapped but placed)
            elif kind == 7: cig = [("S", 5), ("I", 3), ("M", L - 8)]
            else:           cig = [("M", L)]
            flag = 0
            r = rng.integers(0, 40)
            if r == 0: flag |= FLAG_SECONDARY
            if r == 1: flag |= FLAG_DUP
            if r == 2: flag |= FLAG_REVERSE
            if kind == 6: flag |= FLAG_UNMAP
            mapq = int(rng.integers(0, 61))
            qual = rng.integers(2, 41, L).astype(np.uint8)
            if kind == 9: qual[:] = 0xff
            seq_len = 0 if kind == 6 and rng.integers(0, 2) else L
            out.append(encode_read(t, p, mapq, flag, cig, seq_len, qual.tobytes()[:seq_len], name=f"r{i}".encode()))
    return out


def depth_rules(records_raw, tid, n, minq=0, min_baseq=13):
    """load_data_from_bam's accumulation (loaddata.cpp:311-333) with resolve_cigar_pos's positions
    (samfunctions.cpp:38-100), in plain Python: small inputs only."""
    rd = np.zeros(n, dtype=np.int32)
    for rec in records_raw:
        body = rec[4:]
        rtid, pos0, l_nm, mapq, _bin, n_cig, flag, l_seq, _a, _b, _c = struct.unpack_from("<iiBBHHHiiii", body, 0)
        if rtid != tid or pos0 == 0 or mapq < minq or (flag & FLAG_SECONDARY) or (flag & FLAG_DUP):
            continue
        off = 32 + l_nm
        cig = struct.unpack_from(f"<{n_cig}I", body, off)
        qual = body[off + 4 * n_cig + (l_seq + 1) // 2: off + 4 * n_cig + (l_seq + 1) // 2 + l_seq]
        ops = [(c & 0xf, c >> 4) for c in cig]
        anchor = next((k for k, (op, l) in enumerate(ops) if op in (0, 2, 7, 8)), -1)
        if anchor < 0:
            continue
        qop, q = [], 0
        for op, l in ops:
            qop.append(q)
            if op in (0, 1, 4, 7, 8): q += l
        cop = [0] * len(ops)
        end = pos0 + 1
        for k in range(anchor, len(ops)):
            cop[k] = end
            if ops[k][0] in (0, 2, 3, 4): end += ops[k][1]       # M, D, N, S advance; '=' and 'X' do not (reference quirk)
        for k, (op, l) in enumerate(ops):
            if op not in (0, 7): continue
            p1, q1 = cop[k] - 1, qop[k]
            for i in range(l):
                if p1 + i >= n: break
                if qual[q1 + i] >= min_baseq: rd[p1 + i] += 1
    return rd


def reference_depth_dump(ref_bin, libref, bam, fasta_path, chrom, workdir, extra=()):
    """Index the BAM with the reference's own samtools and run the reference with -s: returns its per-base depth."""
    import ctypes
    L = ctypes.CDLL(libref)
    L.bam_index_build.argtypes = [ctypes.c_char_p]
    if L.bam_index_build(os.fsencode(bam)) != 0:
        raise RuntimeError("bam_index_build failed")
    out = os.path.join(workdir, "ref_out.txt")
    # The dump is written right after the pileup.  Later, in -b mode, the reference samples read pairs from a window that
    # starts at 10 Mb (pairrd.cpp:636) and crashes on shorter chromosomes once there are calls to annotate: tolerated here,
    # the caller gets None for the output file then.
    r = subprocess.run([ref_bin, "rsi", "-b", bam, "-f", fasta_path, "-c", chrom, "-o", out, "-s", "-np", *extra],
                       capture_output=True, cwd=workdir, timeout=900)
    dump = out + "." + chrom + "_rd"
    if not os.path.exists(dump):
        raise RuntimeError("the reference did not write its depth dump: " + r.stderr.decode()[-400:])
    a = np.loadtxt(dump, dtype=np.int64)
    return a[:, 1].astype(np.int32), (out if r.returncode == 0 and os.path.exists(out) else None)


def golden_spec():
    return {"refs": [("chrA", 120_011), ("chrS", 200_003)], "coverage": 6.0, "seed": 0xB0B0, "settings": [(0, 13), (20, 0)]}


def build_golden_bam(workdir):
    """The deterministic two-chromosome BAM behind tests/golden/bam_small.npz.  Returns (path, refs, records)."""
    spec = golden_spec()
    recs = []
    for t, (name, n) in enumerate(spec["refs"]):
        for r in synth_reads(n, coverage=spec["coverage"], seed=spec["seed"] + t, ntid=1):
            recs.append(r[:4] + struct.pack("<i", t) + r[8:])
    path = os.path.join(workdir, "golden.bam")
    write_bam(path, spec["refs"], recs, block=20000)
    return path, spec["refs"], recs


def paired_reads_following_depth(depth, n, read_len=100, isize_mean=300, isize_sd=25, seed=3, tid=0, events=(), clipped=0.0, q0=0.0):
    """Properly paired FR reads (both mates present, coordinate sorted) whose local coverage follows `depth`:
    what the reference's read-pair annotation pass expects to find in a BAM."""
    rng = np.random.default_rng(seed)
    L = read_len
    lam = depth[::L].astype(np.float64) / 2.0            # fragments starting per window; two reads per fragment
    items = []
    fid = 0
    for w, k in enumerate(rng.poisson(np.maximum(lam, 0))):
        for p in rng.integers(w * L, min(n - 1, (w + 1) * L), k).tolist():
            isz = max(L + 10, int(rng.normal(isize_mean, isize_sd)))
            p2 = p + isz - L
            if p < 1 or p2 + L >= n:
                continue
            mq = 0 if rng.random() < q0 else int(rng.integers(20, 61))
            q1 = rng.integers(14, 41, L).astype(np.uint8).tobytes()
            q2 = rng.integers(14, 41, L).astype(np.uint8).tobytes()
            nm = f"f{fid}".encode(); fid += 1
            # a share of the reads carries a soft clip: the reference's annotation pass only looks at reads with
            # more than one CIGAR operation (pairrd.cpp:669)
            c1 = [("S", 3), ("M", L - 3)] if rng.random() < clipped else [("M", L)]
            c2 = [("M", L - 3), ("S", 3)] if rng.random() < clipped else [("M", L)]
            items.append((p, encode_read(tid, p, mq, 0x1 | 0x2 | 0x20 | 0x40, c1, L, q1, name=nm, mtid=tid, mpos=p2, tlen=isz)))
            items.append((p2, encode_read(tid, p2, mq, 0x1 | 0x2 | 0x10 | 0x80, c2, L, q2, name=nm, mtid=tid, mpos=p, tlen=-isz)))
    # discordant pairs around the implanted events: spanning pairs for losses, everted pairs for gains
    for (e0, e1, level) in events:
        for k in range(14):
            nm = f"d{fid}".encode(); fid += 1
            qa = rng.integers(14, 41, L).astype(np.uint8).tobytes()
            if level <= 2:     # deletion: forward read ends before the event, its mate starts after it
                pa = int(e0 - L - rng.integers(20, 200)); pb = int(e1 + rng.integers(20, 200))
            else:              # duplication: forward read near the end of the event, its mate near the start (everted)
                pa = int(e1 - L - rng.integers(20, 200)); pb = int(e0 + rng.integers(20, 200))
            if pa < 1 or pb < 1 or max(pa, pb) + L >= n:
                continue
            items.append((pa, encode_read(tid, pa, 40, 0x1 | 0x20 | 0x40, [("M", L - 5), ("S", 5)], L, qa, name=nm, mtid=tid, mpos=pb, tlen=pb + L - pa)))
            items.append((pb, encode_read(tid, pb, 40, 0x1 | 0x10 | 0x80, [("S", 5), ("M", L - 5)], L, qa, name=nm, mtid=tid, mpos=pa, tlen=-(pb + L - pa))))
    items.sort(key=lambda x: x[0])
    return [r for _, r in items]
