"""The GPU BGZF block compressor's algorithm (fqtk_amd/csrc/bgzf_deflate.hpp), no GPU: the phase functions the
HIP kernel runs are executed lane by lane through libfqtk_host.so, and zlib must inflate every payload back to
the input -- FASTQ-like text, every block size around the lane / chunk boundaries, degenerate alphabets (code
length limit, single symbol, no matches), incompressible data (stored block)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from tests import hostlib

MAX_IN = 65280


def deflate(data: bytes, lockstep: int = 1):
    """lockstep=1: the lanes advance token by token, round robin (the GPU's interleaving, approximately);
    lockstep=0: lane after lane."""
    lib = hostlib.lib()
    fn = lib.fqtk_host_bgzf_deflate_emulated
    fn.restype = C.c_int64
    out = (C.c_uint8 * 65536)()
    stored = C.c_int(0)
    n = fn(data, C.c_uint32(len(data)), out, C.c_size_t(65536), C.byref(stored), C.c_int(lockstep))
    assert n > 0
    return bytes(out[:n]), bool(stored.value)


def roundtrip(data: bytes):
    payloads = []
    for lockstep in (0, 1):   # lane after lane, and all lanes step by step (the tables must not care)
        payload, stored = deflate(data, lockstep)
        assert zlib.decompress(payload, -15) == data
        assert len(payload) <= len(data) + 5
        payloads.append(payload)
    assert payloads[0] == payloads[1]
    return len(payload), stored


def fastq_text(n_records, rng, read_len=150, qual=b"I"):
    recs = []
    for i in range(n_records):
        s = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, read_len)])
        q = qual * read_len if len(qual) == 1 else bytes(np.frombuffer(qual, dtype=np.uint8)[rng.integers(0, len(qual), read_len)])
        recs.append(b"@inst:1:FC:1:%010d 1:N:0:ACGTACGT+TTGCAATG\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(recs)


def test_fastq_blocks_round_trip_and_compress():
    rng = np.random.default_rng(1)
    text = fastq_text(400, rng)
    sizes = []
    for off in range(0, len(text) - MAX_IN, MAX_IN):
        n, stored = roundtrip(text[off:off + MAX_IN])
        assert not stored
        sizes.append(n / MAX_IN)
    # constant quality, random bases: ~2 bits per base + cheap headers/qualities -> well under a third
    assert max(sizes) < 0.33, sizes
    # realistic quality strings (a dozen distinct values): still a real compressor, not a stored-block emitter
    text = fastq_text(200, rng, qual=b"FFFFFFFF:,#IIJJ")
    n, stored = roundtrip(text[:MAX_IN])
    assert not stored and n / MAX_IN < 0.6


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 255, 256, 257, 511, 512, 513, 4095, 4096, 65279, 65280])
def test_every_size_around_the_lane_boundaries(n):
    rng = np.random.default_rng(n)
    text = fastq_text(n // 300 + 2, rng)[:n]
    roundtrip(text)
    roundtrip(bytes(rng.integers(0, 256, n, dtype=np.uint8)))       # random bytes
    roundtrip(b"A" * n)                                             # one symbol, long runs
    roundtrip((b"ACGT" * (n // 4 + 1))[:n])                         # period 4


def test_incompressible_data_is_stored_and_degenerate_alphabets_are_valid():
    rng = np.random.default_rng(9)
    n, stored = roundtrip(bytes(rng.integers(0, 256, MAX_IN, dtype=np.uint8)))
    assert stored and n == MAX_IN + 5
    # two symbols, no repeats longer than 3: literals only, a 2-symbol literal code + end of block
    roundtrip(bytes(np.frombuffer(b"AC", dtype=np.uint8)[rng.integers(0, 2, 5000)]))
    # Fibonacci-like counts force the 15-bit length limit (zlib's overflow rule)
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    data = b"".join(bytes([65 + i]) * min(c, 20000) for i, c in enumerate(fib))
    perm = rng.permutation(len(data))
    roundtrip(bytes(np.frombuffer(data, dtype=np.uint8)[perm][:MAX_IN]))
    # all 256 byte values, each once, then a long match-rich tail
    roundtrip(bytes(range(256)) + b"@header:1:2:3 1:N:0\n" * 2000)
    # distances near the 32 KiB window limit
    block = bytes(rng.integers(0, 256, 300, dtype=np.uint8))
    roundtrip(block + bytes(32768 - 300 - 7) + block + bytes(100) + block)


def test_output_is_a_single_final_dynamic_block():
    payload, stored = deflate(fastq_text(150, np.random.default_rng(3))[:40000])
    assert not stored
    assert payload[0] & 1 == 1 and (payload[0] >> 1) & 3 == 2      # BFINAL = 1, BTYPE = 10 (dynamic Huffman)
    d = zlib.decompressobj(-15)
    d.decompress(payload)
    assert d.eof and d.unused_data == b""


def _huffman_lengths(counts, max_bits):
    n = len(counts)
    arr = (C.c_uint32 * n)(*[int(c) for c in counts])
    out = (C.c_uint8 * n)()
    assert hostlib.lib().fqtk_host_huffman_lengths(arr, n, max_bits, out) == 0
    return list(out)


def test_length_limited_codes_are_complete_for_any_counts():
    """The code builder under its length limit (15 bits; 7 for the code-length code): Fibonacci-like counts make
    trees far deeper than the limit, and the repaired code must still be COMPLETE (Kraft sum exactly 1) -- inflate
    rejects an over-subscribed literal/length set.  Regression: the overflow rule once counted only leaves."""
    from fractions import Fraction
    rng = np.random.default_rng(3)
    cases = []
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])
    for n, max_bits in ((286, 15), (30, 15), (19, 7)):
        k = min(n, 40)
        cases.append((fib[:k] + [0] * (n - k), max_bits))                       # deepest possible tree
        cases.append(([0] * (n - k) + fib[:k][::-1], max_bits))
        cases.append(([1] * n, max_bits))
        cases.append(([0] * (n - 1) + [7], max_bits))                             # one symbol: still two codes
        cases.append(([0] * n, max_bits))
        for _ in range(300):
            c = (rng.pareto(0.7, size=n) * 3).astype(np.int64)                    # heavy-tailed, many zeros
            c[rng.random(n) < rng.random()] = 0
            cases.append((np.minimum(c, 1 << 20).tolist(), max_bits))
            g = (2.0 ** rng.integers(0, 22, size=n)).astype(np.int64)             # geometric spread: deep trees
            g[rng.random(n) < 0.5] = 0
            cases.append((g.tolist(), max_bits))
    for counts, max_bits in cases:
        lens = _huffman_lengths(counts, max_bits)
        used = [l for l in lens if l]
        assert len(used) >= 2 and max(used) <= max_bits, (counts, lens)
        assert sum(Fraction(1, 1 << l) for l in used) == 1, (counts, lens)
        for c, l in zip(counts, lens):
            assert (l != 0) or c == 0 or sum(1 for x in counts if x) < 2
            if c:
                assert l != 0


def test_block_with_a_very_deep_literal_tree_inflates():
    """Byte counts 1, 1, 2, 3, 5, 8, ... in one block: the literal code hits the 15-bit limit."""
    fib = [1, 1]
    while sum(fib) < 60000:
        fib.append(fib[-1] + fib[-2])
    rng = np.random.default_rng(5)
    data = np.concatenate([np.full(c, 33 + i, dtype=np.uint8) for i, c in enumerate(fib)])
    rng.shuffle(data)
    data = data.tobytes()[:65280]
    roundtrip(data)


def test_fast_effort_of_low_compression_levels_round_trips_and_is_at_most_a_little_larger():
    """--compression-level 1-3 parse with two match candidates per position instead of three (csrc/bgzf_deflate.hpp:
    effort_of_level): the payload must inflate to the input and stay within a few per cent of the default's."""
    import ctypes as C
    import zlib
    import numpy as np
    from tests import hostlib as H
    fn = H.lib().fqtk_host_bgzf_deflate_level
    fn.restype = C.c_int64
    rng = np.random.default_rng(21)
    recs = []
    for i in range(400):
        recs.append(b"@A00123:45:HXXXXXXXX:1:1101:%d:%d 1:N:0:ACGTACGT+TTGCAATG\n%s\n+\n%s\n" % (
            1000 + 7 * i, int(rng.integers(1000, 30000)), bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)),
            bytes(rng.choice(list(b"FFFFFFFF:,#"), 150).astype(np.uint8))))
    text = b"".join(recs)
    out = (C.c_uint8 * 70000)()
    sizes = {}
    for level in (1, 3, 4, 5, 9, 12):
        tot = 0
        for o in range(0, len(text) - 65280, 65280):
            b = text[o:o + 65280]
            n = fn(b, C.c_uint32(len(b)), out, C.c_size_t(70000), None, 0, level)
            assert n > 0 and zlib.decompress(bytes(out[:n]), -15) == b
            tot += n
        sizes[level] = tot
    assert sizes[1] == sizes[3] and sizes[4] == sizes[5] == sizes[9] == sizes[12]
    assert sizes[5] <= sizes[1] <= sizes[5] * 1.04


def reach_cases(rng):
    """Inputs for bgzf_deflate.hpp phase_reach: the last match of a 64-byte slice runs on (up to 258 bytes: five slices), the
    lanes behind it start where it ends -- whole slices covered, matches cut at their front, rests of one or two bytes."""
    cases = [b"A" * MAX_IN, b"A" * 700, b"AB" * 3000, b"ABC" * 1000 + b"x" + b"ABC" * 1000]
    # runs of every length around the slice and the match maximum, at every phase of the 64-byte grid
    for run in (60, 63, 64, 65, 66, 67, 127, 128, 129, 256, 257, 258, 259, 260, 261, 262, 320, 515, 516, 517, 518):
        for lead in (0, 1, 2, 3, 61, 62, 63):
            filler = bytes(rng.integers(33, 127, 200, dtype=np.uint8))
            cases.append(filler[:lead] + b"Q" * run + filler + b"Q" * run + filler[:77] + b"Q" * (run + 3))
    # a repeated line of every length: matches at distance = line length straddle the grid at every offset
    for line_len in (5, 37, 63, 64, 65, 100, 193, 257, 258, 259, 300):
        line = bytes(rng.integers(65, 91, line_len - 1, dtype=np.uint8)) + b"\n"
        cases.append(line * (4000 // line_len))
    # lines that differ in one or two bytes at random places: matches end and restart within a few bytes of each other
    for _ in range(12):
        line = bytearray(rng.integers(65, 91, int(rng.integers(40, 400)), dtype=np.uint8))
        text = bytearray()
        while len(text) < 20000:
            for _k in range(int(rng.integers(1, 4))):
                line[int(rng.integers(0, len(line)))] = int(rng.integers(65, 91))
            text += line + b"\n"
        cases.append(bytes(text))
    return [c[:MAX_IN] for c in cases]


def test_matches_that_run_past_their_slice_and_the_lanes_they_cover():
    rng = np.random.default_rng(11)
    for data in reach_cases(rng):
        n, stored = roundtrip(data)
        assert not stored
    # and they do what they are for: a 150-byte run of one quality value is one match, not three (0.157 without them)
    assert roundtrip(fastq_text(176, rng)[:MAX_IN])[0] / MAX_IN < 0.145
