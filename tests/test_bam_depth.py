"""BAM pileup -> depth.  CPU: the Python restatement of the reference's rules against the golden
depth the real reference produced (tests/golden/bam_small.npz, tools/make_golden_bam.py).  GPU: the library
(host inflate + device pileup) against the same golden file, against the restatement, and -- where the compiled
reference is present -- against a fresh `-s` dump of the reference binary, plus the command line."""
import os
import subprocess

import numpy as np
import pytest

import bam_util as bu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bam_small.npz")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_rules_match_reference_golden(tmp_path):
    g = np.load(GOLDEN)
    _, refs, recs = bu.build_golden_bam(str(tmp_path))
    for t, (chrom, n) in enumerate(refs):
        for q, Q in bu.golden_spec()["settings"]:
            assert np.array_equal(bu.depth_rules(recs, t, n, minq=q, min_baseq=Q), g[f"{chrom}_q{q}_Q{Q}"]), (chrom, q, Q)


@pytest.mark.gpu
@pytest.mark.parametrize("indexed", [False, True])
def test_library_depth_matches_reference_golden(hotlib, tmp_path, indexed):
    from rsicnv_amd import api
    import oracle
    g = np.load(GOLDEN)
    bam, refs, recs = bu.build_golden_bam(str(tmp_path))
    libref = os.path.join(os.path.dirname(oracle.REF_BIN), "libref.so")
    if indexed:
        if not os.path.exists(libref):
            pytest.skip("no compiled reference to build the .bai with")
        import ctypes
        L = ctypes.CDLL(libref)
        L.bam_index_build.argtypes = [ctypes.c_char_p]
        assert L.bam_index_build(os.fsencode(bam)) == 0
    h = api.RsiHot(0)
    for t, (chrom, n) in enumerate(refs):
        for q, Q in bu.golden_spec()["settings"]:
            st = h.load_depth_bam(bam, chrom, minq=q, min_baseq=Q)
            assert st["n"] == n and st["tid"] == t and st["indexed"] == int(indexed)
            got = h.fetch("depth_in")
            assert np.array_equal(got, g[f"{chrom}_q{q}_Q{Q}"]), (chrom, q, Q, int(np.sum(got != g[f"{chrom}_q{q}_Q{Q}"])))
    h.close()


@pytest.mark.gpu
def test_bam_end_to_end_against_reference_binary(hotlib, tmp_path):
    """A 2 Mb chromosome at 25x through a BAM of many BGZF blocks: depth equal to the reference's -s dump, calls equal
    to the calls from the same depth as arrays; the command line's -s dump equal too.  (The reference binary crashes in its
    annotation pass on chromosomes this short; the whole-file comparison is test_bam_12mb_rows_against_reference_binary.)"""
    import oracle
    from conftest import make_case, calls_equal
    from test_hot_extra import _write_case
    from rsicnv_amd import api
    if not os.path.exists(oracle.REF_BIN):
        pytest.skip("oracle/_ref/rsicnv_ref not built")
    n = 2_000_003
    _, fasta, depth = make_case(hotlib, dict(n=n, seed=0xBA5, model=0, n_events=8, gaps=2, max_len=50000, end_n=8000, gap_len=20000))
    fa, _ = _write_case(str(tmp_path), fasta, depth)
    # properly paired reads whose local coverage follows the synthetic depth (so that there is something to call, and
    # so that the reference's read-pair annotation pass, which runs in -b mode, finds the pairs it samples)
    recs = bu.paired_reads_following_depth(depth, n)
    bam = str(tmp_path / "big.bam")
    bu.write_bam(bam, [("chrS", n)], recs)
    libref = os.path.join(os.path.dirname(oracle.REF_BIN), "libref.so")
    ref_rd, ref_out = bu.reference_depth_dump(oracle.REF_BIN, libref, bam, fa, "chrS", str(tmp_path))
    h = api.RsiHot(0)
    res = h.run_bam(api.make_params(), bam, "chrS", fasta)
    assert res.bam_stats["indexed"] == 1 and res.bam_stats["records"] == len(recs)
    assert np.array_equal(h.fetch("depth_in"), ref_rd)
    res_arr = h.run(api.make_params(), ref_rd, fasta)
    ok, why = calls_equal(res.calls("calls"), res_arr.calls("calls"), rtol=0)
    assert ok, why
    assert len(res.calls("calls")) >= 3
    h.close()
    # command line
    exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
    ours = str(tmp_path / "ours.txt")
    subprocess.run([exe, "rsi", "-b", bam, "-f", fa, "-c", "chrS", "-o", ours, "-np", "-s"], check=True, capture_output=True, timeout=300)
    dump = np.loadtxt(ours + ".chrS_rd", dtype=np.int64)
    assert np.array_equal(dump[:, 1].astype(np.int32), ref_rd)
    def rows(path):
        out = []
        for ln in open(path).read().splitlines():
            cols = ln.split("\t")
            out.append("\t".join(c for i, c in enumerate(cols) if not (len(cols) > 8 and i == 7)))   # drop the RP/Q0 column of data rows
        return out
    if ref_out is not None:   # the reference survives its annotation pass only on chromosomes beyond 10 Mb (see bam_util)
        a, b = rows(ours), rows(ref_out)
        assert a == b, "\n".join(a[:8]) + "\n---\n" + "\n".join(b[:8])
    else:
        assert len(rows(ours)) == 3 + len(res.calls("calls"))


@pytest.mark.gpu
def test_cli_walks_all_chromosomes_of_a_bam(hotlib, tmp_path):
    """-b without -c: every reference of the header with reads is processed (rsi.cpp:2114-2131), depth dumps equal to
    the reference's golden depth, one header in the output file."""
    from conftest import make_case
    g = np.load(GOLDEN)
    bam, refs, _ = bu.build_golden_bam(str(tmp_path))
    fa = str(tmp_path / "ref.fa")
    off = 0
    with open(fa, "wb") as f, open(fa + ".fai", "w") as fai:
        for chrom, n in refs + [("chrEmpty", 50_000)]:
            _, fasta, _ = make_case(hotlib, dict(n=n, seed=0xFA + n, model=0, n_events=1, gaps=0, max_len=3000, end_n=1000))
            head = f">{chrom}\n".encode()
            f.write(head)
            seq = fasta.tobytes()
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + b"\n")
            fai.write(f"{chrom}\t{n}\t{off + len(head)}\t60\t61\n")
            off += len(head) + n + (n + 59) // 60
    exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
    out = str(tmp_path / "all.txt")
    r = subprocess.run([exe, "rsi", "-b", bam, "-f", fa, "-o", out, "-np", "-s"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    log = open(out + ".log").read()
    for chrom, n in refs:
        assert f"#processing {chrom}" in log
        dump = np.loadtxt(out + f".{chrom}_rd", dtype=np.int64)
        assert np.array_equal(dump[:, 1].astype(np.int32), g[f"{chrom}_q0_Q13"]), chrom
    text = open(out).read()
    assert text.count("#CHROM") == 1 and text.startswith(f"#input {bam}\n")
    # -gpus N -workers W: the chromosomes spread over pools (here one device, three in flight), rows still in header
    # order: the file is the same byte for byte (the parallel form of the loop rsi.cpp:2189-2217 and its writer :1594-1608)
    out2 = str(tmp_path / "all_pool.txt")
    r = subprocess.run([exe, "rsi", "-b", bam, "-f", fa, "-o", out2, "-np", "-gpus", "1", "-workers", "3"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert open(out2).read().replace(out2, out) == text.replace(out2, out)
    log2 = open(out2 + ".log").read()
    assert [l for l in log2.splitlines() if l.startswith("#processing")] == [l for l in log.splitlines() if l.startswith("#processing")]
    # -gpus 2 with both logical devices mapped to the one GPU of the box (RSI_HOT_DEVICE_MAP): the partition over devices, a pool
    # per device and the ordered writer give the same file again
    out3 = str(tmp_path / "all_two.txt")
    r = subprocess.run([exe, "rsi", "-b", bam, "-f", fa, "-o", out3, "-np", "-gpus", "2", "-workers", "2"], capture_output=True, timeout=300,
                       env=dict(os.environ, RSI_HOT_DEVICE_MAP="0,0"))
    assert r.returncode == 0, r.stderr.decode()[-800:]
    assert "over 2 device(s)" in r.stderr.decode()
    assert open(out3).read().replace(out3, out) == text.replace(out3, out)


@pytest.mark.gpu
def test_bam_12mb_rows_against_reference_binary(hotlib, tmp_path):
    """a 12 Mb chromosome, where the reference's pair-sampling window (which starts at 10 Mb,
    pairrd.cpp:636) finds reads and the reference binary therefore gets through its annotation pass: the command line's
    output file equals the reference binary's byte for byte -- including the RP / Q0 column, for which the BAM carries
    soft-clipped reads, mapq-0 reads and discordant pairs around the events -- and the depth equals its -s dump."""
    import oracle
    from conftest import make_case
    from test_hot_extra import _write_case
    if not os.path.exists(oracle.REF_BIN):
        pytest.skip("oracle/_ref/rsicnv_ref not built")
    n = 12_000_017
    plan, fasta, depth = make_case(hotlib, dict(n=n, seed=0x5EED0001, model=0, mean=14.0, n_events=5, gaps=1, max_len=20000, end_n=10000, gap_len=30000))
    fa, _ = _write_case(str(tmp_path), fasta, np.zeros(8, dtype=np.int32))
    recs = bu.paired_reads_following_depth(depth, n, seed=11, events=plan["events"], clipped=0.3, q0=0.05)
    bam = str(tmp_path / "c1.bam")
    bu.write_bam(bam, [("chrS", n)], recs)
    libref = os.path.join(os.path.dirname(oracle.REF_BIN), "libref.so")
    ref_rd, ref_out = bu.reference_depth_dump(oracle.REF_BIN, libref, bam, fa, "chrS", str(tmp_path))
    assert ref_out is not None, "the reference binary should survive on a 12 Mb chromosome"
    exe = os.path.join(ROOT, "rsicnv_amd", "bin", "rsicnv")
    ours = str(tmp_path / "ours.txt")
    subprocess.run([exe, "rsi", "-b", bam, "-f", fa, "-c", "chrS", "-o", ours, "-np", "-s"], check=True, capture_output=True, timeout=600)
    dump = np.loadtxt(ours + ".chrS_rd", dtype=np.int64)
    assert np.array_equal(dump[:, 1].astype(np.int32), ref_rd)
    a, b = open(ours).read(), open(ref_out).read()
    assert a == b, a[:1500] + "\n---\n" + b[:1500]
    rows = [ln.split("\t") for ln in a.splitlines() if not ln.startswith("#")]
    assert len(rows) >= 2 and any(not r[7].startswith("RP=0;") for r in rows), "the test BAM should give some call a supporting pair"


@pytest.mark.gpu
@pytest.mark.parametrize("block", [777, 4093, 60000])
def test_records_straddling_bgzf_blocks(hotlib, tmp_path, block):
    """BGZF blocks cut at arbitrary byte positions (records and their length fields span blocks): the loader's
    per-block speculative walk has to fall back to the plain walk wherever a block does not begin on a record."""
    from rsicnv_amd import api
    g = np.load(GOLDEN)
    spec = bu.golden_spec()
    _, refs, recs = bu.build_golden_bam(str(tmp_path))
    bam = str(tmp_path / f"straddle{block}.bam")
    bu.write_bam(bam, spec["refs"], recs, block=block, straddle=True)
    h = api.RsiHot(0)
    for t, (chrom, n) in enumerate(refs):
        st = h.load_depth_bam(bam, chrom)
        assert np.array_equal(h.fetch("depth_in"), g[f"{chrom}_q0_Q13"]), (chrom, block)
        assert st["on_chrom"] == sum(1 for r in recs if int.from_bytes(r[4:8], "little", signed=True) == t)
    h.close()


@pytest.mark.gpu
def test_ingest_error_paths(hotlib, tmp_path):
    """Loaders fail loudly and leave the context usable: missing files, a chromosome the BAM does not have, a file that is
    not BGZF, a BAM cut off in the middle of a block, a chromosome without reads."""
    from rsicnv_amd import api
    bam, refs, recs = bu.build_golden_bam(str(tmp_path))
    chrom, n = refs[0]
    h = api.RsiHot(0)
    with pytest.raises(api.RsiError):
        h.load_depth_bam(str(tmp_path / "nothing.bam"), chrom)
    with pytest.raises(api.RsiError):
        h.load_depth_bam(bam, "chrNotThere")
    junk = tmp_path / "junk.bam"
    junk.write_bytes(b"this is not a BGZF file at all\n" * 100)
    with pytest.raises(api.RsiError):
        h.load_depth_bam(str(junk), chrom)
    cut = tmp_path / "cut.bam"
    raw = open(bam, "rb").read()
    cut.write_bytes(raw[: len(raw) * 2 // 3])
    with pytest.raises(api.RsiError):
        h.load_depth_bam(str(cut), refs[-1][0])
    with pytest.raises(api.RsiError):
        h.load_depth_text(str(tmp_path / "nothing.txt"), 1000)
    # a chromosome of the header that no read maps to: all-zero depth, not an error
    empty = str(tmp_path / "empty.bam")
    import struct
    bu.write_bam(empty, refs, [r for r in recs if struct.unpack("<i", r[4:8])[0] != len(refs) - 1])
    st = h.load_depth_bam(empty, refs[-1][0])
    assert st["n"] == refs[-1][1] and st["used"] == 0
    assert not h.fetch("depth_in").any()
    # and the context still works
    st = h.load_depth_bam(bam, chrom)
    g = np.load(GOLDEN)
    q, Q = 0, 13
    if f"{chrom}_q{q}_Q{Q}" in g:
        assert np.array_equal(h.fetch("depth_in"), g[f"{chrom}_q{q}_Q{Q}"])
    h.close()


@pytest.mark.gpu
def test_malformed_records_are_skipped_not_followed(hotlib, tmp_path):
    """Records a corrupt or crafted BAM could carry: a position of -1 with the chromosome's reference id, a sequence length
    that overruns the record, a CIGAR count that overruns it, a CIGAR longer than the read.  The first is filtered like any
    read the reference's region iterator never yields, the next two are counted as malformed and skipped, the last is cut at
    the read's end; none may touch memory outside the record or the depth array -- the depth equals that of the clean file
    plus the cut read's bases."""
    import struct
    from rsicnv_amd import api
    _, refs, recs = bu.build_golden_bam(str(tmp_path))
    chrom, n = refs[0]
    clean = [r for r in recs if struct.unpack("<i", r[4:8])[0] == 0]
    q = bytes([30] * 50)
    neg = bu.encode_read(0, -1, 60, 0, [("M", 50)], 50, q)
    long_seq = bytearray(bu.encode_read(0, 1000, 60, 0, [("M", 50)], 50, q)); long_seq[20:24] = struct.pack("<i", 1 << 20)        # l_seq
    many_cig = bytearray(bu.encode_read(0, 2000, 60, 0, [("M", 50)], 50, q)); many_cig[16:18] = struct.pack("<H", 60000)        # n_cigar_op
    cig_long = bu.encode_read(0, 3000, 60, 0, [("M", 80)], 50, q)      # 80M on a 50-base read: counts 50 bases
    bad = str(tmp_path / "bad.bam")
    extra = sorted([bytes(long_seq), bytes(many_cig), cig_long], key=lambda r: struct.unpack("<i", r[8:12])[0])
    # keep the file coordinate-sorted: the crafted records go in front of the clean reads that start later
    merged = sorted(clean + extra, key=lambda r: struct.unpack("<i", r[8:12])[0])
    bu.write_bam(bad, refs, [neg] + merged)
    good = str(tmp_path / "good.bam")
    bu.write_bam(good, refs, clean)
    h = api.RsiHot(0)
    h.load_depth_bam(good, chrom)
    want = h.fetch("depth_in").copy()
    want[3000:3050] += 1
    st = h.load_depth_bam(bad, chrom)
    got = h.fetch("depth_in")
    assert st["malformed"] == 2
    assert np.array_equal(got, want)
    h.close()
