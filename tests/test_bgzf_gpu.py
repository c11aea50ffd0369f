"""The BGZF block compressor on the GPU (include/fqtk_bgzf.h), through the C ABI with page-locked host buffers the
kernel reads and writes directly: every payload must inflate (zlib) to its input AND be byte-identical to what the
same phase functions produce lane by lane on the CPU (tests/test_bgzf_deflate.py) -- the algorithm is built to be
independent of how the lanes interleave (min / max tables, lane-private parse state)."""
import ctypes as C
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from fqtk_amd import _lib  # noqa: E402
from tests.test_bgzf_deflate import MAX_IN, deflate as cpu_deflate, fastq_text, reach_cases  # noqa: E402


class Arena:
    def __init__(self, lib, n_blocks):
        self.lib, self.n = lib, n_blocks
        self.pin, self.pout, self.pdesc, self.plen = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert lib.fqtk_pinned_alloc(n_blocks * 65536, C.byref(self.pin)) == 0
        assert lib.fqtk_pinned_alloc(n_blocks * _lib.FQTK_BGZF_OUT_STRIDE, C.byref(self.pout)) == 0
        assert lib.fqtk_pinned_alloc(n_blocks * C.sizeof(_lib.fqtk_bgzf_block), C.byref(self.pdesc)) == 0
        assert lib.fqtk_pinned_alloc(n_blocks * 4, C.byref(self.plen)) == 0
        self.desc = (_lib.fqtk_bgzf_block * n_blocks).from_address(self.pdesc.value)
        self.lens = (C.c_uint32 * n_blocks).from_address(self.plen.value)

    def load(self, blocks):
        for i, b in enumerate(blocks):
            C.memmove(self.pin.value + i * 65536, b, len(b))
            self.desc[i].in_ = self.pin.value + i * 65536
            self.desc[i].out = self.pout.value + i * _lib.FQTK_BGZF_OUT_STRIDE
            self.desc[i].n_in = len(b)
            self.lens[i] = 0

    def payload(self, i):
        return C.string_at(self.pout.value + i * _lib.FQTK_BGZF_OUT_STRIDE, self.lens[i])

    def free(self):
        for p in (self.pin, self.pout, self.pdesc, self.plen):
            self.lib.fqtk_pinned_free(p)


def test_blocks_round_trip_through_the_gpu_compressor():
    lib = _lib.load()
    z = C.c_void_p()
    assert lib.fqtk_bgzf_create(0, C.byref(z)) == 0, lib.fqtk_bgzf_last_error()
    rng = np.random.default_rng(7)
    text = fastq_text(3000, rng, qual=b"FFFFFFFFFF:,#IIJJ<<AA")
    blocks = [text[o:o + MAX_IN] for o in range(0, len(text), MAX_IN)]                       # 15 full FASTQ blocks + a tail
    blocks += [bytes(rng.integers(0, 256, MAX_IN, dtype=np.uint8)), b"A" * MAX_IN, b"A", b"AC", b"ACG", bytes(range(256)),
               (b"ACGT" * 20000)[:MAX_IN], fastq_text(3, rng)[:200], fastq_text(3, rng)[:256], fastq_text(3, rng)[:257]]
    blocks += [fastq_text(300, rng)[:n] for n in (1000, 4096, 4097, 30000, 65279)]
    a = Arena(lib, len(blocks))
    try:
        for rep in range(2):                                                                 # slots 0 and 1, arena reused
            a.load(blocks)
            assert lib.fqtk_bgzf_deflate_enqueue(z, rep, a.pdesc, len(blocks), a.plen) == 0, lib.fqtk_bgzf_last_error()
            assert lib.fqtk_bgzf_wait(z, rep) == 0
            ratios = []
            for i, b in enumerate(blocks):
                p = a.payload(i)
                assert 0 < len(p) <= len(b) + 5
                d = zlib.decompressobj(-15)
                assert d.decompress(p) == b and d.eof and d.unused_data == b"", (i, len(b))
                assert p == cpu_deflate(b)[0], (i, len(b))                                   # bit-exact vs the CPU run of the same phases
                ratios.append(len(p) / len(b))
            assert max(ratios[:15]) < 0.45                                                   # real compression on FASTQ text
        # a slot must be waited on before it is reused; bad arguments are refused
        a.load(blocks[:2])
        assert lib.fqtk_bgzf_deflate_enqueue(z, 2, a.pdesc, 2, a.plen) == 0
        assert lib.fqtk_bgzf_deflate_enqueue(z, 2, a.pdesc, 2, a.plen) == _lib.FQTK_EINVAL
        assert lib.fqtk_bgzf_wait(z, 2) == 0
        assert lib.fqtk_bgzf_deflate_enqueue(z, 9, a.pdesc, 2, a.plen) == _lib.FQTK_EINVAL
    finally:
        a.free()
        lib.fqtk_bgzf_destroy(z)


def test_many_blocks_more_than_workgroups_resident():
    """2000 blocks through one launch: each workgroup loops over several blocks with the same LDS and token scratch."""
    lib = _lib.load()
    z = C.c_void_p()
    assert lib.fqtk_bgzf_create(0, C.byref(z)) == 0
    rng = np.random.default_rng(11)
    text = fastq_text(2000, rng)
    blocks = [text[(37 * i) % 100000:][:int(rng.integers(1, MAX_IN + 1))] for i in range(2000)]
    a = Arena(lib, len(blocks))
    try:
        a.load(blocks)
        assert lib.fqtk_bgzf_deflate_enqueue(z, 0, a.pdesc, len(blocks), a.plen) == 0
        assert lib.fqtk_bgzf_wait(z, 0) == 0
        for i, b in enumerate(blocks):
            assert zlib.decompress(a.payload(i), -15) == b, i
            if i % 50 == 0:
                assert a.payload(i) == cpu_deflate(b)[0], i
    finally:
        a.free()
        lib.fqtk_bgzf_destroy(z)


def test_skewed_alphabets_and_deep_code_trees_on_the_gpu():
    """Blocks built to stress the code builder and the parser: Fibonacci byte counts (code trees far deeper than
    the 15-bit limit; a once over-subscribed code was found this way), geometric counts, a few symbols only, long
    runs with rare interruptions, periodic text at every period up to 40 -- each must inflate and equal the CPU run."""
    lib = _lib.load()
    z = C.c_void_p()
    assert lib.fqtk_bgzf_create(0, C.byref(z)) == 0, lib.fqtk_bgzf_last_error()
    rng = np.random.default_rng(17)
    blocks = []
    fib = [1, 1]
    while sum(fib) < 64000:
        fib.append(fib[-1] + fib[-2])
    for base in (33, 65, 128):
        d = np.concatenate([np.full(c, (base + i) % 256, dtype=np.uint8) for i, c in enumerate(fib)])
        rng.shuffle(d)
        blocks.append(d.tobytes()[:MAX_IN])
        blocks.append(np.sort(d).tobytes()[:MAX_IN])                                          # the same counts as long runs
    for k in (2, 3, 5, 17, 40, 120, 256):
        p = 0.5 ** np.arange(k)
        blocks.append(rng.choice(k, size=MAX_IN, p=p / p.sum()).astype(np.uint8).tobytes())   # geometric counts
        blocks.append(rng.integers(0, k, size=int(rng.integers(1, MAX_IN)), dtype=np.uint8).tobytes())
    for period in range(1, 41):
        unit = bytes(rng.integers(65, 91, period, dtype=np.uint8))
        n = int(rng.integers(period, MAX_IN))
        blocks.append((unit * (n // period + 1))[:n])
    run = np.full(MAX_IN, ord("F"), dtype=np.uint8)
    run[rng.integers(0, MAX_IN, 300)] = rng.integers(33, 75, 300, dtype=np.uint8)
    blocks.append(run.tobytes())
    a = Arena(lib, len(blocks))
    try:
        a.load(blocks)
        assert lib.fqtk_bgzf_deflate_enqueue(z, 0, a.pdesc, len(blocks), a.plen) == 0, lib.fqtk_bgzf_last_error()
        assert lib.fqtk_bgzf_wait(z, 0) == 0
        for i, b in enumerate(blocks):
            p = a.payload(i)
            d = zlib.decompressobj(-15)
            assert d.decompress(p) == b and d.eof and d.unused_data == b"", (i, len(b))
            assert p == cpu_deflate(b)[0], (i, len(b))
    finally:
        a.free()
        lib.fqtk_bgzf_destroy(z)


def test_matches_past_their_slice_on_the_gpu():
    """phase_reach (matches that run on over the next lanes' slices): the crafted runs, repeated lines and nearly equal lines
    of the CPU test, bit-exact against the CPU run of the same phases."""
    lib = _lib.load()
    z = C.c_void_p()
    assert lib.fqtk_bgzf_create(0, C.byref(z)) == 0, lib.fqtk_bgzf_last_error()
    blocks = reach_cases(np.random.default_rng(11))
    a = Arena(lib, len(blocks))
    try:
        a.load(blocks)
        assert lib.fqtk_bgzf_deflate_enqueue(z, 0, a.pdesc, len(blocks), a.plen) == 0, lib.fqtk_bgzf_last_error()
        assert lib.fqtk_bgzf_wait(z, 0) == 0
        for i, b in enumerate(blocks):
            p = a.payload(i)
            d = zlib.decompressobj(-15)
            assert d.decompress(p) == b and d.eof and d.unused_data == b"", (i, len(b))
            assert p == cpu_deflate(b)[0], (i, len(b))
    finally:
        a.free()
        lib.fqtk_bgzf_destroy(z)
