"""The BGZF member decoder (include/fqtk_inflate.h, csrc/bgzf_inflate.hpp).

CPU part: the decoder's own function (inflate_member, written for one 64-lane wavefront) run as 64 fibers by
csrc/host/wave_emu.hpp against zlib -- every level and strategy, stored / fixed / dynamic blocks, several blocks per
member, codes longer than the fast tables' index bits, payloads at every byte alignment, corrupted streams (an error, or
exactly zlib's bytes -- never a crash, never a write past ISIZE).
GPU part: the same members through the C ABI; text, per-member status, newline counts and the CRC check."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

from fqtk_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "fqtk_amd", "lib", "libfqtk_host.so")


def _emu():
    lib = C.CDLL(HOST)
    f = lib.fqtk_host_bgzf_inflate_emulated
    f.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32]
    f.restype = C.c_int
    return f


def raw_deflate(data, level=6, strategy=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return c.compress(data) + c.flush()


def emulated(payload, isize, misalign=0):
    out = C.create_string_buffer(isize + 64)
    guard = b"\xEE" * 64
    C.memmove(C.addressof(out) + isize, guard, 64)
    st = _emu()(payload, len(payload), isize, out, misalign)
    assert out.raw[isize:] == guard                      # nothing is written past ISIZE, whatever the stream says
    return st, out.raw[:isize]


def fastq_text(n, rng, qual=b"FFFFFFFFFF:,#IIJJ<<AA"):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    q = np.frombuffer(qual, dtype=np.uint8)
    recs = []
    for i in range(n):
        L = int(rng.choice([36, 100, 150]))
        recs.append(b"@inst:1:FC:1:%07d:%d 1:N:0:ACGTACGT+TTGCAATG\n%s\n+\n%s\n" % (
            i, int(rng.integers(0, 99999)), acgt[rng.integers(0, 4, L)].tobytes(), q[rng.integers(0, len(q), L)].tobytes()))
    return b"".join(recs)


def members_of_cases(rng, full=True):
    """(text, payload) pairs that cover the decoder's paths (full=False: the large texts at three settings only -- the
    emulated wavefront decodes ~50 KB/s)."""
    geom = bytes(rng.choice(256, 30000, p=np.array([2.0 ** (-i / 9) for i in range(256)]) / sum(2.0 ** (-i / 9) for i in range(256))).astype(np.uint8))
    texts = [b"", b"a", b"hello hello hello hello", b"a" * 1000, bytes(range(256)) * 4, fastq_text(30, rng), fastq_text(400, rng)[:65280 if full else 30000],
             bytes(rng.integers(0, 256, 20000, dtype=np.uint8)), geom if full else geom[:12000], (b"ACGT" * 17000)[:65536], b"\n" * 5000 + b"x"]
    out = []
    for t in texts:
        for level, strategy in ((0, 0), (1, 0), (6, 0), (9, 0), (6, 1), (6, 2), (6, 3), (6, 4)):
            if not full and len(t) > 5000 and (level, strategy) not in ((1, 0), (6, 0), (6, 2)):
                continue
            out.append((t, raw_deflate(t, level, strategy)))
    # several blocks in one member: stored, dynamic and fixed ones mixed
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [fastq_text(30, rng), bytes(rng.integers(0, 256, 500, dtype=np.uint8)), b"abc" * 50, fastq_text(10, rng)]
    comp = b"".join(c.compress(p) + c.flush(zlib.Z_FULL_FLUSH) for p in parts) + c.flush()
    out.append((b"".join(parts), comp))
    return out


def test_emulated_wavefront_inflates_what_zlib_wrote():
    rng = np.random.default_rng(3)
    cases = members_of_cases(rng, full=False)
    # (the big cases once, the small ones at every alignment)
    for k, (text, payload) in enumerate(cases):
        for mis in ((0, 1, 2, 3) if len(text) < 3000 else (k & 3,)):
            st, got = emulated(payload, len(text), mis)
            assert st == 0 and got == text, (k, len(text), mis, st)


def test_emulated_wavefront_rejects_what_zlib_rejects():
    rng = np.random.default_rng(4)
    text = fastq_text(12, rng)
    payload = raw_deflate(text)
    errors = 0
    for bit in range(0, len(payload) * 8, 41):
        bad = bytearray(payload)
        bad[bit >> 3] ^= 1 << (bit & 7)
        st, got = emulated(bytes(bad), len(text))
        d = zlib.decompressobj(-15)
        try:
            ref = d.decompress(bytes(bad))
            zok = d.eof
        except zlib.error:
            ref, zok = None, False
        if st == 0:
            assert zok and ref == got, bit                # accepted: then zlib accepts it too, with the same bytes
        else:
            errors += 1
            assert not (zok and len(ref) == len(text) and d.unused_data == b""), (bit, st)   # rejected: zlib does not return ISIZE bytes cleanly
    assert errors > 20
    assert emulated(payload, len(text) - 1)[0] == 7      # more text than ISIZE
    assert emulated(payload, len(text) + 1)[0] == 9      # less text than ISIZE
    assert emulated(payload[:-3], len(text))[0] in (7, 8)  # truncated payload (what lies behind it decodes to too much, or to nothing)
    assert emulated(b"\x07", 0)[0] == 1                  # reserved block type
    assert emulated(b"\x01\x05\x00\x00\x00", 5)[0] == 2  # stored block whose NLEN is not ~LEN


# ---- GPU ----------------------------------------------------------------------------------------------------------------
def bgzf_member(text, level=6):
    payload = raw_deflate(text, level)
    bsize = 18 + len(payload) + 8 - 1
    head = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize)
    return head + payload + struct.pack("<II", zlib.crc32(text), len(text))


class Batch:
    """Members laid out back to back the way a BGZF file holds them, in page-locked memory."""

    def __init__(self, lib, items):
        self.lib = lib
        file_bytes = bytearray()
        desc = []
        out_off = 0
        for text, payload, crc in items:
            start = len(file_bytes) + 18
            file_bytes += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", min(18 + len(payload) + 7, 65535)) + payload + struct.pack("<II", crc, len(text))
            desc.append((start, out_off, len(payload), len(text), crc))
            out_off += len(text)
        self.n, self.total = len(items), out_off
        self.in_len = len(file_bytes)
        self.ptrs = [C.c_void_p() for _ in range(5)]
        sizes = [self.in_len + 8, max(out_off, 1) + 64, self.n * C.sizeof(_lib.fqtk_inflate_member), self.n * 4, self.n * 4]
        for p, s in zip(self.ptrs, sizes):
            assert lib.fqtk_pinned_alloc(s, C.byref(p)) == 0
        self.pin, self.pout, self.pdesc, self.pstat, self.plines = self.ptrs
        C.memmove(self.pin.value, bytes(file_bytes), self.in_len)
        C.memset(self.pout.value, 0xEE, sizes[1])
        self.desc = (_lib.fqtk_inflate_member * self.n).from_address(self.pdesc.value)
        for i, (a, b, c, d, e) in enumerate(desc):
            self.desc[i].payload_off, self.desc[i].out_off, self.desc[i].payload_len, self.desc[i].isize, self.desc[i].crc = a, b, c, d, e
        self.status = (C.c_uint32 * self.n).from_address(self.pstat.value)
        self.lines = (C.c_uint32 * self.n).from_address(self.plines.value)

    def text(self, i):
        return C.string_at(self.pout.value + self.desc[i].out_off, self.desc[i].isize)

    def free(self):
        for p in self.ptrs:
            self.lib.fqtk_pinned_free(p)


@pytest.mark.gpu
def test_members_inflate_on_the_gpu_to_what_zlib_returns():
    lib = _lib.load()
    z = C.c_void_p()
    assert lib.fqtk_inflate_create(0, C.byref(z)) == 0, lib.fqtk_inflate_last_error()
    rng = np.random.default_rng(3)
    cases = members_of_cases(rng)
    big = fastq_text(30000, rng)
    cases += [(big[o:o + 65280], raw_deflate(big[o:o + 65280], 1 + (o // 65280) % 9)) for o in range(0, len(big), 65280)]
    items = [(t, p, zlib.crc32(t)) for t, p in cases]
    b = Batch(lib, items)
    try:
        for slot in (0, 1):
            C.memset(b.pout.value, 0xEE, b.total + 64)
            assert lib.fqtk_inflate_enqueue(z, slot, b.pin, b.in_len, b.pdesc, b.n, b.pout, b.pstat, b.plines) == 0, lib.fqtk_inflate_last_error()
            assert lib.fqtk_inflate_wait(z, slot) == 0, lib.fqtk_inflate_last_error()
            for i, (t, _, _) in enumerate(items):
                assert b.status[i] == 0, (i, b.status[i], len(t))
                assert b.text(i) == t, (i, len(t))
                assert b.lines[i] == t.count(b"\n"), i
            assert C.string_at(b.pout.value + b.total, 64) == b"\xEE" * 64
        assert lib.fqtk_inflate_enqueue(z, 2, b.pin, b.in_len, b.pdesc, b.n, b.pout, b.pstat, b.plines) == 0
        assert lib.fqtk_inflate_enqueue(z, 2, b.pin, b.in_len, b.pdesc, b.n, b.pout, b.pstat, b.plines) == _lib.FQTK_EINVAL
        assert lib.fqtk_inflate_wait(z, 2) == 0
        assert lib.fqtk_inflate_enqueue(z, 7, b.pin, b.in_len, b.pdesc, b.n, b.pout, b.pstat, b.plines) == _lib.FQTK_EINVAL
    finally:
        b.free()
        lib.fqtk_inflate_destroy(z)


@pytest.mark.gpu
def test_gpu_decoder_reports_bad_members_and_leaves_the_others_alone():
    lib = _lib.load()
    z = C.c_void_p()
    assert lib.fqtk_inflate_create(0, C.byref(z)) == 0
    rng = np.random.default_rng(5)
    text = fastq_text(200, rng)
    payload = raw_deflate(text)
    good = (text, payload, zlib.crc32(text))
    items = [good, (text, payload, zlib.crc32(text) ^ 1), good, (text, payload[:-5], zlib.crc32(text)), good]
    flipped = []
    for bit in range(40, len(payload) * 8, 997):
        bad = bytearray(payload)
        bad[bit >> 3] ^= 1 << (bit & 7)
        flipped.append(bytes(bad))
        items.append((text, bytes(bad), zlib.crc32(text)))
    items.append(good)
    b = Batch(lib, items)
    try:
        assert lib.fqtk_inflate_enqueue(z, 0, b.pin, b.in_len, b.pdesc, b.n, b.pout, b.pstat, b.plines) == 0
        assert lib.fqtk_inflate_wait(z, 0) == 0
        assert [b.status[i] for i in (0, 1, 2, 4)] == [0, _lib.FQTK_INFLATE_ERR_CRC, 0, 0] and b.status[3] in (7, 8)
        for i in (0, 2, 4, b.n - 1):
            assert b.status[i] == 0 and b.text(i) == text and b.lines[i] == text.count(b"\n")
        for k, bad in enumerate(flipped):
            assert b.status[5 + k] != 0, k               # a flipped bit: the stream breaks, or the CRC does
    finally:
        b.free()
        lib.fqtk_inflate_destroy(z)


def _walk(path, max_bytes=1 << 20, max_text=4 << 20, cap=100000):
    lib = C.CDLL(HOST)
    fn = lib.fqtk_host_bgzf_walk
    fn.restype = C.c_int64
    fn.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    rows = (C.c_uint64 * (5 * cap))()
    err = C.create_string_buffer(512)
    n = fn(str(path).encode(), max_bytes, max_text, rows, cap, err, 512)
    if n < 0:
        raise ValueError(err.value.decode())
    return [tuple(rows[5 * i:5 * i + 5]) for i in range(n)]


def test_member_walk_of_the_device_inflate_feeders(tmp_path):
    """csrc/host/bgzf_walk.hpp (what `fqtk demux` hands to fqtk_demuxer_feed): every member of a BGZF file once, in order, with
    the payload / CRC / ISIZE of its header and trailer; runs respect the byte and text limits; files that are not BGZF
    throughout, or damaged, are refused with the reader's messages."""
    rng = np.random.default_rng(9)
    text = fastq_text(800, rng)
    pieces = [text[o:o + 20000] for o in range(0, len(text), 20000)]
    data = b"".join(bgzf_member(p, 1 + i % 9) for i, p in enumerate(pieces)) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    f = tmp_path / "a.fastq.gz"
    f.write_bytes(data)
    rows = _walk(f, max_bytes=20000, max_text=1 << 30)
    assert len(rows) == len(pieces) + 1
    got = b""
    for run, off, plen, isize, crc in rows:
        t = zlib.decompressobj(-15).decompress(data[off:off + plen])
        assert len(t) == isize and zlib.crc32(t) == crc
        got += t
    assert got == text
    runs = {}
    for run, off, plen, isize, crc in rows:
        runs.setdefault(run, []).append((off, plen, isize))
    assert len(runs) >= 3 and sorted(runs) == list(range(len(runs)))
    for r, ms in runs.items():                                   # a run stays below the byte limit unless it is one member
        span = ms[-1][0] + ms[-1][1] + 8 - (ms[0][0] - 18)
        assert span <= 20000 or len(ms) == 1
    small = _walk(f, max_bytes=1 << 30, max_text=45000)          # the text limit: two 20 000-byte members per run
    per_run = {}
    for run, off, plen, isize, crc in small:
        per_run.setdefault(run, []).append(isize)
    assert all(sum(v) <= 45000 or len(v) == 1 for v in per_run.values()) and max(len(v) for v in per_run.values()) >= 2
    # an ordinary gzip member in the middle; a BSIZE that runs past the file; a member of more than 64 KiB of text
    plain = gzip_member = b"\x1f\x8b\x08\x00\0\0\0\0\0\xff" + raw_deflate(b"hello\n") + struct.pack("<II", zlib.crc32(b"hello\n"), 6)
    (tmp_path / "b.gz").write_bytes(bgzf_member(pieces[0]) + plain)
    # (the run ends in front of the ordinary member: the feeder of `fqtk demux` decodes that one as a serial stream; asked for a run
    #  there, the walk says that no BGZF member lies there)
    with pytest.raises(ValueError, match="holds no BGZF member at byte %d" % len(bgzf_member(pieces[0]))):
        _walk(tmp_path / "b.gz")
    (tmp_path / "c.gz").write_bytes(data[:len(data) // 2])
    with pytest.raises(ValueError, match="bad BGZF block size|holds no BGZF member"):
        _walk(tmp_path / "c.gz")
    big = bytearray(bgzf_member(pieces[0]))
    big[-4:] = struct.pack("<I", 70000)
    (tmp_path / "d.gz").write_bytes(bytes(big))
    with pytest.raises(ValueError, match="more than 64 KiB"):
        _walk(tmp_path / "d.gz")


def test_gzip_member_headers_as_the_serial_gzip_feeders_skip_them():
    """bgzf_walk.hpp gzip_header_len (RFC 1952 2.3): FEXTRA, FNAME, FCOMMENT and FHCRC in every combination; what is no gzip
    header, reserved flags and truncated headers give 0."""
    lib = C.CDLL(HOST)
    fn = lib.fqtk_host_gzip_header_len
    fn.restype = C.c_uint64
    fn.argtypes = [C.c_char_p, C.c_size_t]
    body = raw_deflate(b"ACGT\n" * 10) + struct.pack("<II", zlib.crc32(b"ACGT\n" * 10), 50)
    for flg in range(0, 32):
        if flg & 1:                       # FTEXT changes nothing
            pass
        head = bytes([0x1f, 0x8b, 8, flg, 0, 0, 0, 0, 0, 3])
        if flg & 4:
            head += struct.pack("<H", 5) + b"extra"
        if flg & 8:
            head += b"name.fastq\0"
        if flg & 16:
            head += b"a comment\0"
        if flg & 2:
            head += b"\x12\x34"
        data = head + body
        assert fn(data, len(data)) == len(head), flg
        assert zlib.decompressobj(-15).decompress(data[len(head):-8]) == b"ACGT\n" * 10
    ok = bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]) + body
    assert fn(ok, len(ok)) == 10
    assert fn(b"@read\nACGT\n+\nIIII\n" * 3, 54) == 0                      # plain text
    assert fn(bytes([0x1f, 0x8b, 8, 0x20]) + ok[4:], len(ok)) == 0           # a reserved flag
    assert fn(bytes([0x1f, 0x8b, 9]) + ok[3:], len(ok)) == 0                 # not DEFLATE
    trunc = bytes([0x1f, 0x8b, 8, 8, 0, 0, 0, 0, 0, 3]) + b"a name without its end" * 2
    assert fn(trunc, len(trunc)) == 0
