"""CPU tests of the pieces the serial-gzip device path of `fqtk demux` stands on (the reference reads any gzip file through one
gz-aware reader: /root/reference/src/bin/commands/demux.rs:844-849):
  * find_block_start (csrc/bgzf_inflate.hpp), the device's block-start search, run on the wave emulator against the host's own search
    (host/parallel_gunzip.hpp) and against block boundaries known from zlib's flush points;
  * the stream mode of the device decoder reporting the last block boundary it reached when it runs out of room;
  * RegionInflate (host/region_inflate.hpp), the sequential decoder a stretch falls back to, against zlib."""
import ctypes as C
import zlib

import numpy as np
import pytest

from tests import hostlib as H


def fastq_text(rng, n, qual_levels=40):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for i in range(n):
        L = int(rng.integers(30, 151))
        seq = acgt[rng.integers(0, 4, L)].tobytes()
        q = (33 + rng.integers(0, qual_levels, L)).astype(np.uint8).tobytes()
        out.append(b"@inst:1:FC:1:%d:%d:%d 1:N:0:0\n%s\n+\n%s\n" % (i // 1000, i, int(rng.integers(0, 99999)), seq, q))
    return b"".join(out)


def stream_with_known_boundaries(text, level, part=40000):
    """Raw DEFLATE of text; a sync flush behind every `part` bytes: the block after the flush's empty stored block starts at a known bit."""
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp, bounds = b"", []
    for o in range(0, len(text), part):
        comp += c.compress(text[o:o + part])
        if o + part < len(text):
            comp += c.flush(zlib.Z_SYNC_FLUSH)
            bounds.append(len(comp) * 8)
    comp += c.flush()
    return comp, bounds


def find_emulated(comp, lo, hi, text_only=False):
    f = H.lib().fqtk_host_find_block_start_emulated
    f.restype = C.c_uint64
    return f(comp, C.c_uint32(len(comp)), C.c_uint32(lo), C.c_uint32(hi), C.c_int(1 if text_only else 0))


def find_host(comp, lo, hi):
    f = H.lib().fqtk_host_find_block_start
    f.restype = C.c_uint64
    return f(comp, C.c_size_t(len(comp)), C.c_uint64(lo), C.c_uint64(hi))


NONE = (1 << 64) - 1


@pytest.mark.parametrize("level", [1, 6, 9])
def test_block_start_search_of_the_device_finds_the_known_boundaries_and_agrees_with_the_host_search(level):
    rng = np.random.default_rng(100 + level)
    text = fastq_text(rng, 1500)
    comp, bounds = stream_with_known_boundaries(text, level)
    assert len(bounds) >= 4
    pad = comp + bytes(2048)   # (the host search keeps 1 KiB clear of the data's end)
    for b in bounds[:-1]:      # (the last part may end the stream with a FINAL block, which is no chunk start)
        assert find_emulated(pad, b - 41, b + 64) == b, (level, b)
        assert find_emulated(pad, b - 41, b + 64, text_only=True) == b   # (FASTQ is 7-bit text: the stricter search finds it too)
        assert find_emulated(pad, b + 1, b + 40) == NONE          # bits inside a header are no header
        assert find_emulated(pad, b - 300, b) in (NONE,)           # nothing in the flush marker / the end of the block before it
    # anywhere: the same answer as the host's search over the same range
    agree = found = 0
    for lo in rng.integers(0, max(1, len(comp) * 8 - 70000), 6):
        lo = int(lo)
        e, h = find_emulated(pad, lo, lo + 60000), find_host(pad, lo, lo + 60000)
        assert e == h, (level, lo, e, h)
        agree += 1
        found += e != NONE
    assert agree == 6 and found >= 1


def test_block_start_search_rejects_noise_and_short_data():
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, 6000, dtype=np.uint8).tobytes() + bytes(2048)
    assert find_emulated(noise, 0, 40000) == find_host(noise, 0, 40000)
    assert find_emulated(bytes(64), 0, 512) == NONE
    # a block of 8-bit data: a start for the plain search, none for the one that insists on 7-bit text
    blob = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes() * 3
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(b"x" * 10) + c.flush(zlib.Z_SYNC_FLUSH)
    b = len(comp) * 8
    comp += c.compress(blob) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(b"tail") + c.flush() + bytes(2048)
    assert find_emulated(comp, b - 20, b + 64) == b and find_emulated(comp, b - 20, b + 64, text_only=True) == NONE
    assert find_emulated(b"\x05" * 3, 0, 24) == NONE          # BTYPE = 2 in every byte, nothing behind it


def stream_emulated(comp, start_bit, stop_bit, cap):
    f = H.lib().fqtk_host_inflate_stream_emulated
    sym = (C.c_uint16 * (cap + 600))()
    res = (C.c_uint32 * 4)()
    st = f(comp, C.c_uint32(len(comp)), C.c_uint32(start_bit), C.c_uint32(stop_bit), sym, C.c_uint32(cap), res)
    return st, list(res), sym


def test_a_chunk_that_runs_out_of_room_reports_the_last_block_boundary_it_reached():
    rng = np.random.default_rng(7)
    text = fastq_text(rng, 700)
    comp, bounds = stream_with_known_boundaries(text, 6, part=12000)
    assert len(bounds) >= 5
    pad = comp + bytes(64)
    # room for the whole stream: status 0, ends with the final block
    st, res, sym = stream_emulated(pad, 0, 0xFFFFFFFF, len(text) + 100)
    assert st == 0 and res[0] == len(text) and res[2] == 1 and res[3] >= len(bounds)
    assert bytes(sym[:len(text)]) == text                      # (no window in front of bit 0: every symbol is a byte)
    # room for two and a half parts: FQTK_INFLATE_ERR_OUTPUT, and what it reports is a true boundary with the text up to it
    st, res, sym = stream_emulated(pad, 0, 0xFFFFFFFF, 30000)
    assert st == 7 and res[3] >= 2 and res[2] == 0
    n, end_bit = res[0], res[1]
    assert 0 < n <= 30000 and bytes(sym[:n]) == text[:n]
    d = zlib.decompressobj(-15)
    assert d.decompress(comp[:(end_bit + 7) // 8] if end_bit % 8 == 0 else comp)[:n] == text[:n]
    # ... from which a decoder that has the window carries on: the rest decodes to the rest of the text
    rest = region_inflate(comp, end_bit, text[max(0, n - 32768):n], len(comp) * 8, 1 << 30)
    assert rest[0] == text[n:] and rest[2] == 1


def region_inflate(data, from_bit, window, until_bit, max_text, cap=1 << 24):
    f = H.lib().fqtk_host_region_inflate
    out = C.create_string_buffer(cap)
    res = (C.c_uint64 * 3)()
    wa = C.create_string_buffer(32768)
    err = C.create_string_buffer(256)
    w = None
    if window is not None:
        w = bytes(32768 - len(window)) + window
    rc = f(data, C.c_size_t(len(data)), C.c_uint64(from_bit), w, C.c_uint64(until_bit), C.c_uint64(max_text), out, C.c_size_t(cap), res, wa, err, C.c_size_t(256))
    if rc != 0:
        raise ValueError(err.value.decode())
    return out.raw[:res[0]], res[1], res[2], wa.raw


@pytest.mark.parametrize("level", [0, 1, 6])
def test_region_inflate_decodes_stretches_of_a_stream_from_known_boundaries_like_zlib(level):
    rng = np.random.default_rng(40 + level)
    text = fastq_text(rng, 2500)
    comp, bounds = stream_with_known_boundaries(text, level, part=50000)
    # from the start to the first boundary at or behind a bit
    got, end_bit, final, wa = region_inflate(comp, 0, None, bounds[1], 1 << 30)
    assert final == 0 and end_bit >= bounds[1] and text.startswith(got) and len(got) >= 100000
    assert wa == text[len(got) - 32768:len(got)]
    # on from there with the window it left, to the end
    got2, end2, final2, _ = region_inflate(comp, end_bit, wa, len(comp) * 8, 1 << 30)
    assert final2 == 1 and got + got2 == text and end2 <= len(comp) * 8
    # max_text cuts at a block boundary: never mid-block, always progress
    at, pieces, window = 0, [], None
    for _ in range(1000):
        g, at, fin, window = region_inflate(comp, at, window, len(comp) * 8, 20000)
        pieces.append(g)
        window = window if len(b"".join(pieces)) >= 32768 else b"".join(pieces)
        if fin:
            break
    assert b"".join(pieces) == text and len(pieces) >= 3
    # a whole gzip file: the member's first block follows the header
    gz = b"\x1f\x8b\x08\x00\0\0\0\0\0\x03" + comp + zlib.crc32(text).to_bytes(4, "little") + (len(text) & 0xFFFFFFFF).to_bytes(4, "little")
    g, e, fin, _ = region_inflate(gz, 80, None, len(gz) * 8, 1 << 30)
    assert g == text and fin == 1 and (e + 7) // 8 == len(gz) - 8


def test_region_inflate_reports_corruption_and_streams_that_end_too_soon():
    rng = np.random.default_rng(9)
    text = fastq_text(rng, 400)
    comp = zlib.compress(text, 6)[2:-4]
    with pytest.raises(ValueError):
        region_inflate(comp[:len(comp) // 2], 0, None, 1 << 40, 1 << 30)
    bad = bytearray(comp)
    errors = 0
    for bit in range(100, len(comp) * 8, 997):
        bad[bit // 8] ^= 1 << (bit % 8)
        try:
            got = region_inflate(bytes(bad), 0, None, 1 << 40, 1 << 30)
            d = zlib.decompressobj(-15)
            try:
                ref = d.decompress(bytes(bad))
                assert (not d.eof) or ref == got[0]
            except zlib.error:
                pass   # (a stream zlib rejects later than this decoder stops: the caller's CRC check is what counts)
        except ValueError:
            errors += 1
        bad[bit // 8] ^= 1 << (bit % 8)
    assert errors > 5
