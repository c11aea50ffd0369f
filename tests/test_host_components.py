"""CPU tests of the C++ host components either side of the hot path, through the
ctypes shim: read structures, header rewriting (the reference's own vectors, demux.rs:2084-2196),
FASTQ parsing, BGZF, metrics, and the CLI validation that happens before any GPU work."""
import gzip
import math
import os
import stat
import struct
import zlib

import pytest

from tests import hostlib as H


# ---- read structures (read-structure 0.2.0; pinned by demux.rs call sites/tests) ---------------------
@pytest.mark.parametrize("text", ["17B100T", "10M8B7C100T", "7B+T", "8B100T", "9B100T", "8B", "100T", "9B",
                                  "4B4M8S", "4B100T", "100S3B", "6B1S1M1T", "17B20T20S20T20S20T", "+T", "+B",
                                  "+M", "7B", "16C8B126T", "150T"])
def test_read_structure_round_trips(text):
    canon, _, segs = H.read_structure(text)
    assert canon == text
    off = 0
    for o, l, _ in segs:
        assert o == off
        off += max(l, 0)


def test_read_structure_min_length_and_kinds():
    # demux.rs:298: sum of fixed lengths + 1 per variable segment
    assert H.read_structure("8B92T")[1] == 100
    assert H.read_structure("+T")[1] == 1
    assert H.read_structure("7B+T")[1] == 8
    assert H.read_structure("10M8B7C100T")[2] == [(0, 10, "M"), (10, 8, "B"), (18, 7, "C"), (25, 100, "T")]
    assert H.read_structure("7B+T")[2] == [(0, 7, "B"), (7, -1, "T")]
    assert H.read_structure(" 8b 92t ")[0] == "8B92T"          # whitespace / case tolerated (fgbio behaviour)


@pytest.mark.parametrize("bad", ["", "8", "B", "8X", "0T", "+T8B", "8B+", "++T", "8BT", "-8B"])
def test_read_structure_rejects_malformed(bad):
    with pytest.raises(ValueError):
        H.read_structure(bad)


def test_segment_spans():
    assert H.segment_spans("10M8B7C100T", 125) == [(0, 10), (10, 18), (18, 25), (25, 125)]
    assert H.segment_spans("7B+T", 12) == [(0, 7), (7, 12)]
    assert H.segment_spans("8B", 20) == [(0, 8)]               # bases past a fixed structure are ignored
    assert H.segment_spans("4B4M8S", 16) == [(0, 4), (4, 8), (8, 16)]


# ---- header rewriting: the reference's own vectors (demux.rs:2084-2196) -----------------------------
def test_write_header_reference_vectors():
    b, umi = ["ACGT", "GGTT"], ["AACCGGTT"]
    assert H.write_header(1, "inst:123:ABCDE:1:204:1022:2108 1:N:0:0", b, []) == \
        "@inst:123:ABCDE:1:204:1022:2108 1:N:0:ACGT+GGTT"
    assert H.write_header(2, "inst:123:ABCDE:1:204:1022:2108 1:Y:0:0", b, umi) == \
        "@inst:123:ABCDE:1:204:1022:2108:AACCGGTT 2:Y:0:ACGT+GGTT"
    assert H.write_header(2, "inst:123:ABCDE:1:204:1022:2108:AAAA 1:Y:0:TTTT", b, umi) == \
        "@inst:123:ABCDE:1:204:1022:2108:AAAA+AACCGGTT 2:Y:0:TTTT+ACGT+GGTT"
    assert H.write_header(1, "q1", b, umi) == "@q1:AACCGGTT 1:N:0:ACGT+GGTT"
    assert H.write_header(1, "q1 0:0", b, umi) == "@q1:AACCGGTT 0:0:ACGT+GGTT"
    with pytest.raises(ValueError, match="8 segments"):
        H.write_header(1, "q1:1:2:3:4:5:6:7:8:9:10", b, umi)


def test_write_header_more_cases():
    assert H.write_header(1, "ex_0", ["AAAAAAAAGATTACAGA"], []) == "@ex_0 1:N:0:AAAAAAAAGATTACAGA"   # demux.rs:1327
    assert H.write_header(1, "ex_0", ["AAAAAAAA"], ["ATCGATCGAT"]) == "@ex_0:ATCGATCGAT 1:N:0:AAAAAAAA"  # :1383
    assert H.write_header(3, "q 1:N:0:ACGT", ["TT"], ["AA", "CC"]) == "@q:AA+CC 3:N:0:ACGT+TT"
    assert H.write_header(1, "q 1:N:0:", ["TT"], []) == "@q 1:N:0:TT"
    assert H.write_header(1, "a:b:c:d:e:f:g:h:i", ["TT"], []) == "@a:b:c:d:e:f:g:h:i 1:N:0:TT"   # no UMI: name untouched
    with pytest.raises(ValueError, match="4 segments"):
        H.write_header(1, "q 1:N:0:A:B", ["TT"], [])


# ---- FASTQ parsing ------------------------------------------------------------------------------------
def test_parse_fastq_plain_gz_multimember_crlf_and_empty_reads(tmp_path):
    recs = [("r0 1:N:0:1", "ACGT", "IIII"), ("r1", "", ""), ("r2", "NNNNNNNN", "########")]
    text = "".join(f"@{h}\n{s}\n+\n{q}\n" for h, s, q in recs)
    p = tmp_path / "a.fastq"
    p.write_text(text)
    assert H.parse_fastq(p) == recs
    assert H.parse_fastq(p, batch=1) == recs
    g = tmp_path / "a.fastq.gz"
    with open(g, "wb") as fh:                       # two gzip members
        fh.write(gzip.compress(text[: len(text) // 2].encode()))
        fh.write(gzip.compress(text[len(text) // 2:].encode()))
    assert H.parse_fastq(g) == recs
    b = tmp_path / "b.fastq.gz"
    b.write_bytes(H.bgzf(text.encode()))            # our own BGZF output is readable too
    assert H.parse_fastq(b) == recs
    c = tmp_path / "c.fastq"
    c.write_bytes(text.replace("\n", "\r\n").encode())
    assert H.parse_fastq(c) == recs
    d = tmp_path / "d.fastq"
    d.write_text(text.rstrip("\n"))                 # no trailing newline
    assert H.parse_fastq(d) == recs
    assert H.parse_fastq(tmp_path / "a.fastq", batch=2) == recs
    e = tmp_path / "e.fastq"
    e.write_text("")
    assert H.parse_fastq(e) == []


@pytest.mark.parametrize("text", ["r0\nACGT\n+\nIIII\n", "@r0\nACGT\n-\nIIII\n", "@r0\nACGT\n+\nIII\n", "@r0\nACGT\n+\n"])
def test_parse_fastq_rejects_malformed(tmp_path, text):
    p = tmp_path / "bad.fastq"
    p.write_text(text)
    with pytest.raises(ValueError, match="Unexpected error parsing FASTQs"):
        H.parse_fastq(p)


# ---- BGZF ---------------------------------------------------------------------------------------------
def test_bgzf_is_valid_multi_member_gzip_with_bc_fields_and_eof_marker():
    import numpy as np
    rng = np.random.default_rng(0)
    data = bytes(rng.choice(np.frombuffer(b"ACGT\n@+I", dtype=np.uint8), size=300_000))
    blob = H.bgzf(data, level=5)
    assert gzip.decompress(blob) == data
    # walk the blocks: 'BC' extra field, BSIZE, ISIZE <= 64 KiB; last block is the 28-byte EOF marker
    off, total, nblocks = 0, 0, 0
    while off < len(blob):
        assert blob[off:off + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", blob, off + 10)[0]
        assert xlen == 6 and blob[off + 12:off + 14] == b"BC"
        bsize = struct.unpack_from("<H", blob, off + 16)[0] + 1
        isize = struct.unpack_from("<I", blob, off + bsize - 4)[0]
        crc = struct.unpack_from("<I", blob, off + bsize - 8)[0]
        payload = zlib.decompress(blob[off + 18:off + bsize - 8], -15)
        assert len(payload) == isize <= 65536 and zlib.crc32(payload) == crc
        total += isize
        off += bsize
        nblocks += 1
    assert total == len(data) and nblocks == math.ceil(len(data) / 65280) + 1
    assert blob[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    assert H.bgzf(b"") == blob[-28:]


# ---- metrics ------------------------------------------------------------------------------------------
def test_metrics_follow_demux_metric_update():
    frac, to_mean, to_best = H.metrics([30, 10, 0, 60])      # 3 samples + unmatched
    assert frac == [0.3, 0.1, 0.0, 0.6]
    mean = 40 / 3
    assert to_mean == [30 / mean, 10 / mean, 0.0, 60 / mean]
    assert to_best == [1.0, 10 / 30, 0.0, 2.0]
    f, m, b = H.metrics([0, 0])                              # nothing demultiplexed: 0/0
    assert all(math.isnan(x) for x in f + m + b)


def test_format_f64_is_shortest_round_trip():
    for v, s in [(1.0, "1.0"), (0.5, "0.5"), (0.1, "0.1"), (1 / 3, "0.3333333333333333"), (0.0, "0.0"),
                 (2.0, "2.0"), (1e-7, "1e-7"), (123456.0, "123456.0"), (float("nan"), "NaN"), (float("inf"), "inf")]:
        assert H.format_f64(v) == s
    for v in (0.30000000000000004, 7.123456789012345e-5, 12345.678901234567):
        assert float(H.format_f64(v)) == v


def test_load_samples_matches_sample_group_rules(tmp_path):
    p = tmp_path / "m.tsv"
    p.write_text("sample_id\tbarcode\ns1\tGATTACA\ns2\tCATGCTA\n\n")
    assert H.load_samples(p) == 2
    for body, msg in [("sample\tbarcode\ns1\tGATTACA\n", "header mismatch"),
                      ("sample_id\tbarcode\ns1\tGATTACA\ns1\tCATGCTA\n", "Each sample name must be unique"),
                      ("sample_id\tbarcode\ns1\tGATTACA\ns2\tGATTACA\n", "Each sample barcode must be unique"),
                      ("sample_id\tbarcode\ns1\tGATTACA\ns2\tGATTAC\n", "All barcodes must have the same length"),
                      ("sample_id\tbarcode\ns1\tgattaca\n", "All sample barcode bases must be one of"),
                      ("sample_id\tbarcode\n", "Must provide one or more sample"),
                      ("\n", "Must provide one or more sample"),
                      ("sample_id,barcode\ns1,GATTACA\n", "header mismatch")]:
        p.write_text(body)
        with pytest.raises(ValueError, match=msg):
            H.load_samples(p)
    with pytest.raises(ValueError, match=r"No such file or directory \(os error 2\)"):
        H.load_samples(tmp_path / "missing.tsv")


# ---- CLI validation that fails before any GPU work (demux.rs:1137-1290,1879-1981) ---------------------
def _basic(tmp_path):
    r1 = H.fastq_file(tmp_path, "read1", "ex", ["GATTACA"])
    i1 = H.fastq_file(tmp_path, "index1", "ex", ["GATTGGG"])
    meta = H.metadata_file(tmp_path, ["GATTGGG"])
    return r1, i1, meta


def test_cli_different_number_of_read_structures_and_inputs_fails(tmp_path):
    r1, i1, meta = _basic(tmp_path)
    for rs in (["+T"], ["+T", "+B", "+T"]):
        r = H.run_demux([r1, i1], rs, meta, tmp_path / "out")
        assert r.returncode != 0
        assert "The same number of read structures should be given as FASTQs" in r.stderr
        assert f"{len(rs)} read-structures provided for 2 FASTQs" in r.stderr


def test_cli_read_only_output_dir_fails(tmp_path):
    r1, i1, meta = _basic(tmp_path)
    out = tmp_path / "ro"
    out.mkdir()
    os.chmod(out, stat.S_IRUSR | stat.S_IXUSR | stat.S_IRGRP | stat.S_IXGRP)
    r = H.run_demux([r1, i1], ["+T", "+B"], meta, out)
    os.chmod(out, 0o755)
    assert r.returncode != 0 and "cannot be read-only" in r.stderr


def test_cli_missing_input_and_too_few_threads_are_reported_together(tmp_path):
    r1, i1, meta = _basic(tmp_path)
    r = H.run_demux([r1, str(tmp_path / "nope.fastq")], ["+T", "+B"], meta, tmp_path / "out", threads=2)
    assert r.returncode != 0
    assert "doesn't exist" in r.stderr and "Threads provided 2 was too low! Must be 5 or more." in r.stderr
    assert "The following errors with the input(s) were detected:" in r.stderr


def test_cli_rejects_bad_flags_before_touching_anything(tmp_path):
    r1, i1, meta = _basic(tmp_path)
    r = H.run_demux([r1, i1], ["+T", "8Q"], meta, tmp_path / "out")
    assert r.returncode != 0 and "unknown type" in r.stderr
    r = H.run_demux([r1, i1], ["+T", "+B"], meta, tmp_path / "out", skip_reasons=["because"])
    assert r.returncode != 0 and "Invalid skip reason: because" in r.stderr
    r = H.run_demux([r1, i1], ["+T", "+B"], meta, tmp_path / "out", output_types=["Q"])
    assert r.returncode != 0 and "Error parsing segment types to report" in r.stderr
    r = H.run_demux([r1, i1], ["+T", "+B"], meta, tmp_path / "out", max_mismatches=300)
    assert r.returncode != 0 and "out of range integral type conversion" in r.stderr


def test_large_inputs_plain_gzip_multimember_and_bgzf_block_parallel_agree(tmp_path):
    """The three byte sources of the FASTQ reader (read(2), zlib stream, BGZF blocks inflated in parallel by
    helper threads) deliver the same records; a corrupt or truncated BGZF member is an error, not silence."""
    import gzip
    import numpy as np
    rng = np.random.default_rng(3)
    n = 120_000
    seqs = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=(n, 60))]
    text = b"".join(b"@r%d 1:N:0:AC\n%s\n+\n%s\n" % (i, seqs[i].tobytes(), b"I" * 60) for i in range(n))
    plain = tmp_path / "big.fastq"
    plain.write_bytes(text)
    gz = tmp_path / "big.fastq.gz"
    gz.write_bytes(gzip.compress(text, 1))
    multi = tmp_path / "multi.fastq.gz"
    cut = [0, len(text) // 3, len(text) // 2, len(text)]
    multi.write_bytes(b"".join(gzip.compress(text[a:b], 1) for a, b in zip(cut, cut[1:])))
    bg = tmp_path / "big.bgzf.fastq.gz"
    blob = H.bgzf(text, 1)
    bg.write_bytes(blob)
    want = H.fastq_digest(plain)                       # regular file: memory-mapped, parsed in place
    assert want[0] == n and want[2] == 0
    os.environ["FQTK_NO_MMAP"] = "1"                   # the read(2) + producer-thread path a pipe would take
    try:
        assert H.fastq_digest(plain) == want and H.fastq_digest(plain, batch=777) == want
    finally:
        del os.environ["FQTK_NO_MMAP"]
    assert H.fastq_digest(plain, batch=777) == want
    assert H.fastq_digest(gz)[:2] == want[:2] and H.fastq_digest(gz)[2] == 1
    assert H.fastq_digest(multi)[:2] == want[:2]
    for helpers in (0, 1, 3):
        for batch in (1000, 50_000):
            got = H.fastq_digest(bg, batch=batch, helpers=helpers)
            assert got[:2] == want[:2] and got[2] == 2, (helpers, batch)
    bad = bytearray(blob)
    bad[len(blob) // 2] ^= 0x55                       # somewhere inside a deflate payload / trailer
    (tmp_path / "bad.gz").write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        H.fastq_digest(tmp_path / "bad.gz")
    (tmp_path / "cut.gz").write_bytes(blob[:len(blob) // 2])
    with pytest.raises(ValueError):
        H.fastq_digest(tmp_path / "cut.gz")


# ---- the streaming inflate of single-stream gzip inputs (fast_inflate.hpp) against zlib --------------
def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def _fastq_like(n, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        L = int(rng.integers(30, 160))
        seq = "".join(rng.choice(list("ACGTN"), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        qual = "".join(rng.choice(list("FFFFFF:,#"), size=L))
        recs.append(f"@inst:7:FC:1:{1000 + i % 97}:{rng.integers(1, 30000)}:{rng.integers(1, 30000)} 1:N:0:ACGT\n{seq}\n+\n{qual}\n")
    return "".join(recs).encode()


@pytest.mark.parametrize("level", [0, 1, 2, 4, 6, 9])
@pytest.mark.parametrize("strategy", [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
def test_gunzip_every_level_and_strategy(level, strategy):
    """Stored, fixed and dynamic blocks, long and short codes, long runs (distance 1), far matches."""
    import numpy as np
    rng = np.random.default_rng(level * 10 + strategy)
    parts = [_fastq_like(3000, 5), bytes(rng.integers(0, 256, 70000, dtype=np.uint8)), b"A" * 100000,
             bytes(rng.integers(0, 4, 50000, dtype=np.uint8)), b"abcdefg" * 9000, _fastq_like(500, 6) * 3]
    data = b"".join(parts)
    assert H.gunzip(_gz(data, level, strategy)) == data


def test_gunzip_spans_many_pieces_and_windows():
    """12 MB of output: three 4 MiB pieces, matches that reach back across piece boundaries."""
    data = _fastq_like(40000, 1)
    data = data * (12_000_000 // len(data) + 1)
    for level in (1, 6):
        assert H.gunzip(_gz(data, level)) == data


def test_gunzip_headers_members_and_tiny_streams():
    data = _fastq_like(200, 2)
    one = gzip.compress(data, mtime=0)
    assert H.gunzip(one) == data
    assert H.gunzip(one + gzip.compress(b"") + one) == data + data            # concatenated members, an empty one
    assert H.gunzip(gzip.compress(b"")) == b""
    assert H.gunzip(gzip.compress(b"x")) == b"x"
    # FEXTRA + FNAME + FCOMMENT + FHCRC set by hand
    raw = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = raw.compress(data) + raw.flush()
    hdr = bytes([0x1f, 0x8b, 8, 4 | 8 | 16 | 2, 0, 0, 0, 0, 0, 3]) + struct.pack("<H", 5) + b"extra" + b"name\0" + b"comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    trailer = struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)
    assert H.gunzip(hdr + body + trailer) == data
    assert H.gunzip(one + b"\0" * 100) == data                                # trailing padding is ignored, as gzread does
    for memlevel in (1, 9):                                                    # small / large symbol buffers: block sizes
        assert H.gunzip(_gz(data * 20, 6, memlevel=memlevel)) == data * 20
    for wbits in (25, 28):                                                     # 512-byte and 4-KiB windows
        assert H.gunzip(_gz(data, 9, wbits=wbits)) == data


def test_gunzip_rejects_corrupt_and_truncated_streams_without_crashing():
    import numpy as np
    data = _fastq_like(3000, 3)
    good = _gz(data, 6)
    with pytest.raises(ValueError):
        H.gunzip(good[:-1])                                                    # trailer cut
    with pytest.raises(ValueError):
        H.gunzip(good[:len(good) // 2])                                        # stream cut
    bad = bytearray(good)
    bad[-5] ^= 1                                                               # CRC
    with pytest.raises(ValueError):
        H.gunzip(bytes(bad))
    rng = np.random.default_rng(4)
    survived = 0
    for _ in range(300):                                                       # random damage: an error or (rarely) the
        b = bytearray(good)                                                    # CRC catches it; never a crash
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(10, len(b)))] = int(rng.integers(0, 256))
        cut = int(rng.integers(20, len(b) + 1)) if rng.random() < 0.3 else len(b)
        try:
            out = H.gunzip(bytes(b[:cut]), cap=len(data) * 4 + (1 << 20))
            survived += out == data
        except (ValueError, AssertionError):
            pass
    assert survived <= 300


def test_single_stream_gzip_input_through_the_reader_uses_the_fast_decoder(tmp_path, monkeypatch):
    data = _fastq_like(30000, 8)
    p = tmp_path / "x.fq.gz"
    p.write_bytes(_gz(data, 6))
    plain = tmp_path / "x.fq"
    plain.write_bytes(data)
    want = H.fastq_digest(plain)
    got = H.fastq_digest(p)
    assert got[2] == 1 and got[:2] == want[:2]
    monkeypatch.setenv("FQTK_ZLIB_INFLATE", "1")                               # zlib's gzread path stays available
    assert H.fastq_digest(p)[:2] == want[:2]


# ---- one gzip stream decoded by several threads (parallel_gunzip.hpp) against zlib ------------------------
@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FILTERED), (6, zlib.Z_RLE), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)])
@pytest.mark.parametrize("threads,chunk", [(2, 65536), (5, 30000), (8, 200000)])
def test_parallel_gunzip_every_level_and_strategy(level, strategy, threads, chunk):
    """Small chunks, so that a few MB cross hundreds of chunk boundaries: every accepted chunk must have been decoded
    exactly (chain of block starts), everything else must fall back to the sequential decoder -- same bytes either way.
    Fixed-Huffman and stored streams have no dynamic block to find: the whole file falls back."""
    import numpy as np
    rng = np.random.default_rng(level * 7 + strategy)
    data = _fastq_like(9000, 11) + bytes(rng.integers(0, 4, 200000, dtype=np.uint8)) + b"A" * 300000 + _fastq_like(4000, 12)
    out, rounds, fallbacks = H.gunzip_parallel(_gz(data, level, strategy), threads=threads, chunk=chunk, cap=len(data) + 1000)
    assert out == data
    if level >= 1 and strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED):
        assert rounds >= (1 if chunk >= 200000 else 4) and fallbacks < rounds          # the parallel path really ran


def test_parallel_gunzip_members_windows_and_damage():
    import numpy as np
    data = _fastq_like(20000, 13)
    one = _gz(data, 6)
    many = one + gzip.compress(b"") + _gz(data[:100000], 1) + one                # members end inside stretches
    want = data + data[:100000] + data
    for threads, chunk in ((3, 50000), (6, 20000)):
        out, rounds, _ = H.gunzip_parallel(many, threads=threads, chunk=chunk, cap=len(want) + 1000)
        assert out == want and rounds > 3
    far = bytes(np.random.default_rng(1).integers(0, 256, 30000, dtype=np.uint8))
    data2 = (far + _fastq_like(300, 14)) * 40                                     # matches that reach ~32 KiB back, across chunks
    out, rounds, _ = H.gunzip_parallel(_gz(data2, 9), threads=4, chunk=40000, cap=len(data2) + 1000)
    assert out == data2 and rounds > 1
    rng = np.random.default_rng(2)
    for _ in range(60):                                                           # damage: an error, never a crash, never wrong bytes accepted silently
        b = bytearray(one)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(10, len(b)))] = int(rng.integers(0, 256))
        cut = int(rng.integers(20, len(b) + 1)) if rng.random() < 0.3 else len(b)
        try:
            out, _, _ = H.gunzip_parallel(bytes(b[:cut]), threads=4, chunk=30000, cap=len(data) * 3)
            assert out == data                                                    # (only if the damage missed everything that matters)
        except ValueError:
            pass


def test_reader_decodes_a_gzip_input_with_several_threads(tmp_path, monkeypatch):
    data = _fastq_like(40000, 15)
    p = tmp_path / "y.fq.gz"
    p.write_bytes(_gz(data, 6))
    plain = tmp_path / "y.fq"
    plain.write_bytes(data)
    want = H.fastq_digest(plain)
    monkeypatch.setenv("FQTK_GZ_CHUNK", "30000")            # 3 MB of gzip = a hundred chunks
    for threads in ("2", "5"):
        monkeypatch.setenv("FQTK_GZ_THREADS", threads)
        got = H.fastq_digest(p, batch=7777)
        assert got[2] == 1 and got[:2] == want[:2]
    monkeypatch.setenv("FQTK_GZ_THREADS", "1")              # the sequential decoder
    assert H.fastq_digest(p)[:2] == want[:2]


def test_records_longer_than_a_piece_s_head_room(tmp_path, monkeypatch):
    """Reads of 70-300 kb: a record that straddles two pieces is longer than the 64 KiB of head room, so the reader
    joins the two pieces instead (and a record can span several 4 MiB pieces of a BGZF or gzip input)."""
    import numpy as np
    rng = np.random.default_rng(21)
    recs = []
    for i in range(120):
        L = int(rng.integers(70_000, 300_000))
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), L).tobytes()
        recs.append(b"@long%d some comment\n" % i + seq + b"\n+\n" + b"F" * L + b"\n")
    data = b"".join(recs)
    plain = tmp_path / "l.fq"
    plain.write_bytes(data)
    want = H.fastq_digest(plain, batch=7)
    assert want[0] == 120
    monkeypatch.setenv("FQTK_NO_MMAP", "1")
    assert H.fastq_digest(plain, batch=7)[:2] == want[:2]
    monkeypatch.delenv("FQTK_NO_MMAP")
    gz = tmp_path / "l.fq.gz"
    gz.write_bytes(_gz(data, 6))
    assert H.fastq_digest(gz, batch=5)[:2] == want[:2]
    monkeypatch.setenv("FQTK_GZ_CHUNK", "50000")
    monkeypatch.setenv("FQTK_GZ_THREADS", "3")
    assert H.fastq_digest(gz, batch=11)[:2] == want[:2]
    bg = tmp_path / "l.bgzf.fq.gz"
    bg.write_bytes(H.bgzf(data))
    assert H.fastq_digest(bg, batch=3)[:2] == want[:2]


# ---- members / stored blocks that end inside the decoder's last 64 bytes ------------------------
def test_gunzip_small_members_and_stored_blocks_near_the_end_of_the_file():
    """The decoder switches to a padded copy of the file's last 64 bytes; its bit buffer may still hold bytes from
    before the switch.  A trailer or a stored-block header right behind the switch point used to be read from the
    wrong place ('CRC mismatch' on valid files)."""
    import numpy as np
    rng = np.random.default_rng(11)
    big = (b"ACGTACGTTTGA" * 420)[:5000]
    for tail_len in list(range(0, 40)) + [60, 100, 200]:
        tail = bytes(rng.choice(list(b"ACGT"), tail_len).astype(np.uint8))
        for lvl in (1, 6, 9):
            m1, m2 = _gz(big, lvl), _gz(tail, lvl)
            assert 18 <= len(m2) <= 250
            want = big + tail
            assert H.gunzip(m1 + m2) == want
            assert H.gunzip(m1 + m2 + m2) == want + tail
            got, _, _ = H.gunzip_parallel(m1 + m2, threads=3, chunk=20_000)
            assert got == want
    # stored blocks (level 0) whose headers land near the end
    for n in (0, 1, 5, 30, 50, 58, 59, 60, 61, 70):
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert H.gunzip(_gz(big, 6) + _gz(d, 0)) == big + d
        # a deflate stream that ends with an empty stored block (Z_SYNC_FLUSH) then the final block
        c = zlib.compressobj(6, zlib.DEFLATED, 31)
        body = c.compress(big) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(d) + c.flush(zlib.Z_FULL_FLUSH) + c.flush()
        assert H.gunzip(body) == big + d


def test_gunzip_fuzz_against_zlib_many_member_shapes():
    import numpy as np
    rng = np.random.default_rng(5)
    for _ in range(60):
        members, want = [], b""
        for _ in range(int(rng.integers(1, 5))):
            n = int(rng.choice([0, 1, 7, 40, 300, 5000, 70_000]))
            d = bytes(rng.choice(list(b"ACGTN\n@+I"), n).astype(np.uint8))
            members.append(_gz(d, int(rng.integers(0, 10))))
            want += d
        blob = b"".join(members)
        assert H.gunzip(blob) == want
        assert zlib.decompressobj(31).decompress(members[0]) == want[:len(zlib.decompress(members[0], 31))]


def _bgzf_members(data, level=6, block=60_000):
    out = []
    for i in range(0, max(len(data), 1), block):
        d = data[i:i + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(d) + c.flush()
        bsize = 18 + len(body) + 8
        out.append(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", bsize - 1) +
                   body + struct.pack("<II", zlib.crc32(d), len(d)))
    return out


def test_bgzf_reader_checks_crc_and_isize_and_takes_a_plain_member_behind_bgzf_ones(tmp_path):
    text = b"".join(b"@r%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(20_000))
    members = _bgzf_members(text)
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    good = tmp_path / "good.fq.gz"
    good.write_bytes(b"".join(members) + eof)
    n, dig, kind = H.fastq_digest(good)
    assert (n, kind) == (20_000, 2)
    # a forged CRC field is an error (python's gzip and the reference's flate2 reader reject the same file)
    m = bytearray(members[1])
    m[-8] ^= 0x55
    bad = tmp_path / "badcrc.fq.gz"
    bad.write_bytes(members[0] + bytes(m) + b"".join(members[2:]) + eof)
    with pytest.raises(ValueError, match="corrupt BGZF block"):
        H.fastq_digest(bad)
    with pytest.raises(Exception):
        gzip.decompress(bad.read_bytes())
    # ISIZE beyond what a BGZF member may hold: an error, not a 4 GiB allocation
    m = bytearray(members[1])
    m[-4:] = struct.pack("<I", 0xFFFFFFF0)
    bad.write_bytes(members[0] + bytes(m) + eof)
    with pytest.raises(ValueError, match="64 KiB"):
        H.fastq_digest(bad)
    # BGZF members followed by an ordinary gzip member: one multi-member stream to gzread and to the reference
    more = b"".join(b"@s%d\nTTTTGGGGCC\n+\nIIIIIIIIII\n" % i for i in range(3000))
    mixed = tmp_path / "mixed.fq.gz"
    mixed.write_bytes(b"".join(members) + gzip.compress(more, mtime=0))
    plain = tmp_path / "all.fq"
    plain.write_bytes(text + more)
    assert gzip.decompress(mixed.read_bytes()) == text + more
    n2, dig2, kind2 = H.fastq_digest(mixed)
    n3, dig3, _ = H.fastq_digest(plain)
    assert (n2, dig2, kind2) == (n3, dig3, 2) and n2 == 23_000


def test_reader_teardown_after_a_parse_error_mid_file_with_parallel_decoders(tmp_path, monkeypatch):
    """The decoder threads read the mapping of the compressed file: it must outlive them."""
    recs = [b"@r%d\nACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIII\n" % i for i in range(120_000)]
    recs[60_000] = b"@broken\nACGT\n+\nII\n"
    p = tmp_path / "broken.fq.gz"
    p.write_bytes(gzip.compress(b"".join(recs), 1, mtime=0))
    monkeypatch.setenv("FQTK_GZ_THREADS", "4")
    monkeypatch.setenv("FQTK_GZ_CHUNK", "65536")
    for _ in range(5):
        with pytest.raises(ValueError, match="lengths differ"):
            H.fastq_digest(p, batch=1000)


@pytest.mark.timeout(120)
def test_a_damaged_gzip_input_is_an_error_for_everyone_who_asks_not_a_wait_for_ever(tmp_path):
    """Round 5 (found by the damaged-stream runs of tests/test_cli_bgzf_inputs_gpu.py): `fqtk demux` judges the record size of every
    input by its first MiB before the reader threads start; a decoder that fails reports its error ONCE, as the last piece of its
    queue -- the estimate took it, and the reader waited on an empty queue for ever.  The error now stays with the source."""
    import ctypes as C
    import gzip
    text = "".join(f"@r{i} x\nACGTACGTAC\n+\nFFFFFFFFFF\n" for i in range(20000)).encode()
    good = gzip.compress(text, 6)
    fn = H.lib().fqtk_host_estimate_then_read
    fn.restype = C.c_int64
    def run(data):
        p = tmp_path / "x.fastq.gz"
        p.write_bytes(data)
        est = C.c_uint64(0)
        err = C.create_string_buffer(512)
        n = fn(str(p).encode(), C.c_uint64(1000), C.byref(est), err, C.c_size_t(512))
        return n, est.value, err.value.decode()
    n, est, err = run(good)
    assert n == 1000 and est > 1000 * 20 and err == ""
    for at, bit in ((3, 2), (11, 3), (40, 4), (len(good) // 2, 0)):
        bad = bytearray(good)
        bad[at] ^= 1 << bit
        n, est, err = run(bytes(bad))
        assert (n == -1 and "gzip" in err) or n == 1000, (at, n, err)       # damaged at the start: an error, at once; further in: the first records still come
    bad = bytearray(good)
    bad[3] ^= 4
    assert run(bytes(bad))[0] == -1
