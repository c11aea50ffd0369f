"""CPU checks of what the GPU record pipeline (include/fqtk_demux.h) shares with the host: the record as a list of
pieces (csrc/record_format.hpp) against host/header.hpp and the reference's own header vectors
(/root/reference/src/bin/commands/demux.rs:2084-2196), the newline-counting reader (FastqSource::next_raw) against the
files' bytes, and the lane-parallel CRC-32 of the BGZF kernel against zlib."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

from tests import hostlib as H


def format_record(header, read_num, bsegs, msegs, bases, quals):
    fn = H.lib().fqtk_host_format_record
    fn.restype = C.c_int64
    out = C.create_string_buffer(1 << 16)
    b = (C.c_char_p * max(len(bsegs), 1))(*[s.encode() for s in bsegs])
    m = (C.c_char_p * max(len(msegs), 1))(*[s.encode() for s in msegs])
    n = fn(header.encode(), C.c_uint32(read_num), b, C.c_uint32(len(bsegs)), m, C.c_uint32(len(msegs)), bases.encode(),
           quals.encode(), out, C.c_size_t(1 << 16))
    if n < 0:
        raise ValueError(int(n))
    return out.raw[:n].decode()




def test_pieces_equal_write_header_on_the_reference_vectors_and_more():
    cases = [
        ("q1", 1, [], []), ("q1", 2, ["ACGT"], []), ("q1", 1, ["ACGT", "TTTT"], ["GG"]), ("q1", 12, ["A"], ["C", "G", "T"]),
        ("inst:1:fc:2:3:4:5 1:N:0:ATCACG", 2, ["ACGT"], []), ("inst:1:fc:2:3:4:5 1:N:0:ATCACG", 1, ["ACGT", "TT"], ["GG"]),
        ("inst:1:fc:2:3:4:5 1:N:0:0", 1, ["ACGT"], []), ("inst:1:fc:2:3:4:5 1:N:0:", 1, ["ACGT"], []),
        ("inst:1:fc:2:3:4:5:UMI 2:Y:18:ATC", 1, ["AC"], ["GGG"]), ("a:b:c:d:e:f:g:h 1:N:0:1", 3, [], ["AC", "GT"]),
        ("name comment", 1, ["ACGT"], []), ("name a:b", 1, ["ACGT"], []), ("name a:b:", 1, ["ACGT"], ["T"]), ("name a:b:c", 7, [], []),
        ("x 1:N:0:5", 10, ["AAAA", "CCCC", "GGGG"], []), ("x y z 1:N:0:ACG", 1, ["T"], []),
    ]
    for header, rn, b, m in cases:
        want = H.write_header(rn, header, b, m) + "\nACGTTGCA\n+\nIIIIFFFF\n"
        assert format_record(header, rn, b, m, "ACGTTGCA", "IIIIFFFF") == want, (header, rn, b, m)
    # empty segment: still four lines
    assert format_record("q", 1, ["AC"], [], "", "") == H.write_header(1, "q", ["AC"], []) + "\n\n+\n\n"


@pytest.mark.parametrize("header, msegs, code", [
    ("a:b:c:d:e:f:g:h:i 1:N:0:A", ["AC"], 1),   # more than 8 name segments (only with a UMI to place)
    ("q1 ", [], 2),                              # empty comment
    ("q1 1:N:0:A:B", [], 3),                     # five comment fields
    ("q1 1:::", [], None),                       # three colons, nothing after the first but colons: fine
    ("q1 :::5", [], None),
])
def test_header_errors_are_the_ones_write_header_raises(header, msegs, code):
    try:
        want = H.write_header(1, header, ["AC"], msegs)
    except ValueError:
        want = None
    if want is None:
        with pytest.raises(ValueError) as e:
            format_record(header, 1, ["AC"], msegs, "A", "I")
        if code is not None:
            assert e.value.args[0] == -1 - code
    else:
        assert format_record(header, 1, ["AC"], msegs, "A", "I") == want + "\nA\n+\nI\n"


def test_random_headers_agree_with_write_header():
    rng = np.random.default_rng(3)
    alphabet = list("abAC:: +019N")
    agree = errors = 0
    for _ in range(4000):
        header = "".join(rng.choice(alphabet, int(rng.integers(1, 40))))
        if "\0" in header:
            continue
        b = ["ACGT"[: int(rng.integers(1, 5))] for _ in range(int(rng.integers(0, 4)))]
        m = ["TTGA"[: int(rng.integers(1, 5))] for _ in range(int(rng.integers(0, 3)))]
        rn = int(rng.choice([1, 2, 9, 10, 123]))
        try:
            want = H.write_header(rn, header, b, m) + "\nAC\n+\nII\n"
        except ValueError:
            want = None
        try:
            got = format_record(header, rn, b, m, "AC", "II")
        except ValueError:
            got = None
        assert got == want, (header, rn, b, m)
        agree += want is not None
        errors += want is None
    assert agree > 1000 and errors > 100


def read_raw(path, batch, cuts=False):
    fn = {False: H.lib().fqtk_host_read_raw, True: H.lib().fqtk_host_read_cuts, "assisted": H.lib().fqtk_host_read_cuts_assisted}[cuts]
    fn.restype = C.c_int64
    cap = 64 << 20
    out = np.empty(cap, dtype=np.uint8)
    counts = (C.c_uint64 * 100000)()
    n_out = C.c_size_t()
    err = C.create_string_buffer(512)
    n = fn(str(path).encode(), C.c_uint64(batch), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.byref(n_out), counts,
           C.c_size_t(100000), err, C.c_size_t(512))
    if n < 0:
        raise ValueError(err.value.decode())
    return out[: n_out.value].tobytes(), [int(counts[i]) for i in range(n)]


def test_next_raw_cuts_every_kind_of_input_by_counting_lines(tmp_path, monkeypatch):
    recs = [b"@r%d some:comment\n%s\n+\n%s\n" % (i, b"ACGT" * (1 + i % 40), b"IIII" * (1 + i % 40)) for i in range(30_000)]
    text = b"".join(recs)
    plain = tmp_path / "a.fq"
    plain.write_bytes(text)
    gz = tmp_path / "a.fq.gz"
    gz.write_bytes(gzip.compress(text, 1, mtime=0))
    bg = tmp_path / "b.fq.gz"
    bg.write_bytes(H.bgzf(text))
    for path in (plain, gz, bg):
        for batch in (1, 7, 1000, 29_999, 30_000, 50_000):
            if batch == 1 and path is not plain:
                continue
            got, counts = read_raw(path, batch)
            assert got == text
            assert sum(counts) == 30_000 and all(c == batch for c in counts[:-1]) and 0 < counts[-1] <= batch
    monkeypatch.setenv("FQTK_NO_MMAP", "1")      # a plain input read in pieces (what a pipe gets)
    got, counts = read_raw(plain, 4096)
    assert got == text and sum(counts) == 30_000
    monkeypatch.delenv("FQTK_NO_MMAP")
    # the end of the input: no final newline, trailing blank lines, CRLF -- and a file cut short
    (tmp_path / "c.fq").write_bytes(text[:-1])
    assert read_raw(tmp_path / "c.fq", 1000)[0] == text
    (tmp_path / "d.fq").write_bytes(text + b"\n\r\n\n")
    assert read_raw(tmp_path / "d.fq", 1000)[0] == text
    (tmp_path / "e.fq").write_bytes(text + b"@cut\nACGT\n")
    with pytest.raises(ValueError, match="truncated record"):
        read_raw(tmp_path / "e.fq", 1000)
    (tmp_path / "f.fq").write_bytes(b"")
    assert read_raw(tmp_path / "f.fq", 10) == (b"", [])


def test_next_cut_hands_out_the_same_chunks_without_copying(tmp_path):
    """A large plain input is cut by a thread that only counts newlines (FastqSource::next_cut) and copied by others."""
    recs = [b"@r%d c\n%s\n+\n%s\n" % (i, b"ACGT" * (1 + i % 30), b"IIII" * (1 + i % 30)) for i in range(20_000)]
    text = b"".join(recs)
    for name, data in (("a.fq", text), ("b.fq", text[:-1]), ("c.fq", text + b"\n\r\n")):
        p = tmp_path / name
        p.write_bytes(data)
        for batch in (1, 999, 20_000, 30_000):
            if batch == 1 and name != "a.fq":
                continue
            assert read_raw(p, batch, cuts=True) == read_raw(p, batch) and read_raw(p, batch, cuts=True)[0] == text
    (tmp_path / "d.fq").write_bytes(text + b"@cut\nAC\n")
    with pytest.raises(ValueError, match="truncated record"):
        read_raw(tmp_path / "d.fq", 1000, cuts=True)


def test_a_cut_stays_mapped_until_its_copier_hands_it_back(tmp_path):
    """the cutter unmapped everything a fixed distance behind it, but the copier of `fqtk demux` may be four cuts
    behind (one in hand, two queued, one just finished) and a cut may be hundreds of MiB.  Unmapping now follows
    release_cut(): the consumer here stays `hold` cuts behind while the source unmaps in 64 KiB steps; with the old rule
    the copies below would read unmapped pages."""
    fn = H.lib().fqtk_host_read_cuts_held
    fn.restype = C.c_int64
    recs = [b"@r%d c\n%s\n+\n%s\n" % (i, b"ACGT" * (1 + i % 30), b"IIII" * (1 + i % 30)) for i in range(60_000)]
    text = b"".join(recs)
    p = tmp_path / "a.fq"
    p.write_bytes(text)
    for batch, hold in ((1000, 4), (5000, 3), (700, 9), (60_000, 2)):
        cap = len(text) + 16
        out = np.empty(cap, dtype=np.uint8)
        counts = (C.c_uint64 * 1000)()
        n_out, unmapped = C.c_size_t(), C.c_size_t()
        err = C.create_string_buffer(512)
        n = fn(str(p).encode(), C.c_uint64(batch), C.c_uint64(hold), C.c_uint64(65536), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap),
               C.byref(n_out), counts, C.c_size_t(1000), C.byref(unmapped), err, C.c_size_t(512))
        assert n >= 0, err.value
        assert out[: n_out.value].tobytes() == text and sum(counts[i] for i in range(n)) == 60_000
        if n > hold + 3:   # something was unmapped behind the consumer, and never past what it had handed back
            held_bytes = sum(len(b"".join(recs[k * batch:(k + 1) * batch])) for k in range(n - hold, n))
            assert 0 < unmapped.value <= len(text) - held_bytes + 65536


def test_a_second_thread_counting_for_the_cutter_changes_nothing(tmp_path):
    """FastqSource::attach_count_assistant: the later 256 KiB steps of a cut are counted by a second thread (it guesses the
    cut's length from the last one): cuts of 4-40 MB whose lengths drift up and down, so the guess overshoots and falls short."""
    rng = np.random.default_rng(5)
    recs = []
    for i in range(120_000):
        n = 40 + 25 * ((i // 9000) % 7) + int(rng.integers(0, 9))   # record lengths drift in bands
        recs.append(b"@r%d\n%s\n+\n%s\n" % (i, b"A" * n, b"I" * n))
    text = b"".join(recs)
    assert 24 << 20 < len(text) < 60 << 20
    for name, data in (("a.fq", text), ("b.fq", text[:-1])):
        p = tmp_path / name
        p.write_bytes(data)
        for batch in (11_000, 25_000, 50_001, 200_000):
            plain = read_raw(p, batch, cuts=True)
            assert read_raw(p, batch, cuts="assisted") == plain and plain[0] == text
            assert plain[1] == [batch] * (120_000 // batch) + ([120_000 % batch] if 120_000 % batch else [])


def test_count_newlines_every_alignment_and_length():
    # through next_raw: lines of every length put the cut at every offset inside the 128-byte SIMD step
    for n in range(1, 300, 7):
        rec = b"@" + b"h" * n + b"\n" + b"A" * (n % 13) + b"\n+\n" + b"I" * (n % 13) + b"\n"
        text = rec * 50
        p = "/tmp/fqtk_nl_%d.fq" % os.getpid()
        with open(p, "wb") as fh:
            fh.write(text)
        try:
            got, counts = read_raw(p, 3)
            assert got == text and counts == [3] * 16 + [2]
        finally:
            os.unlink(p)


def test_lane_parallel_crc_equals_zlib():
    fn = H.lib().fqtk_host_bgzf_crc_emulated
    fn.restype = C.c_uint32
    rng = np.random.default_rng(9)
    for n in [1, 2, 127, 128, 129, 255, 256, 257, 1000, 4096, 65279, 65280] + [int(x) for x in rng.integers(1, 65281, 12)]:
        d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert fn(d, C.c_uint32(n)) == zlib.crc32(d), n
