"""CPU-side checks of the drop-in boundary: the C-ABI library loads (no GPU needed) and exports every
symbol include/fqtk_match.h declares; argument validation that happens before any device work."""
import ctypes as C
import os
import re

import pytest

from fqtk_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    """Every function any header under include/ declares (fqtk_match.h: the matcher; fqtk_bgzf.h: the BGZF compressor)."""
    syms = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not h.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        syms |= set(re.findall(r"\b(fqtk_[a-z0-9_]+)\s*\(", src))
    return sorted(syms)


def test_header_declares_expected_entry_points():
    syms = _declared_symbols()
    for must in ["fqtk_matcher_create", "fqtk_matcher_assign_batch", "fqtk_matcher_assign_batch_device",
                 "fqtk_matcher_assign1", "fqtk_matcher_destroy", "fqtk_last_error", "fqtk_pinned_alloc",
                 "fqtk_matcher_enqueue", "fqtk_matcher_wait", "fqtk_matcher_counts", "fqtk_matchers_allreduce_counts",
                 "fqtk_bgzf_create", "fqtk_bgzf_deflate_enqueue", "fqtk_bgzf_wait", "fqtk_bgzf_destroy"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} declared under include/ but not exported"


def test_binding_covers_every_declared_symbol():
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert bound == set(_declared_symbols())
    _lib.load()


def test_match_struct_is_4_bytes_little_endian_layout():
    assert C.sizeof(_lib.fqtk_match_t) == 4
    m = _lib.fqtk_match_t(0x1234, 5, 7)
    assert bytes(m) == bytes([0x34, 0x12, 5, 7])


def test_abi_version_and_device_count_do_not_need_a_gpu():
    lib = _lib.load()
    assert lib.fqtk_abi_version() == 5
    n = C.c_int(-1)
    assert lib.fqtk_device_count(C.byref(n)) == _lib.FQTK_OK
    assert n.value >= 0


def _create(barcodes, L=None, mm=1, delta=2, device=0):
    lib = _lib.load()
    arr = (C.c_char_p * max(len(barcodes), 1))(*[b.encode() for b in barcodes])
    h = C.c_void_p()
    rc = lib.fqtk_matcher_create(arr, len(barcodes), len(barcodes[0]) if L is None and barcodes else (L or 0),
                                 mm, delta, device, C.byref(h))
    return rc, h, _lib.last_error()


def test_create_rejects_bad_tables_before_touching_the_device():
    # messages follow barcode_matching.rs:61-65 and samples.rs:117-122
    rc, _, msg = _create([])
    assert rc == _lib.FQTK_EINVAL and "Must provide at least one sample" in msg
    rc, _, msg = _create(["ACGT", ""], L=4)
    assert rc == _lib.FQTK_EINVAL and "cannot be empty" in msg
    rc, _, msg = _create(["ACGT", "ACG"], L=4)
    assert rc == _lib.FQTK_EINVAL and "same length" in msg
    rc, _, msg = _create(["A" * 129])
    assert rc == _lib.FQTK_EINVAL and "128" in msg


def test_no_cpu_fallback_without_a_device():
    lib = _lib.load()
    n = C.c_int(0)
    lib.fqtk_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    rc, _, msg = _create(["ACGT", "TTTT"])
    assert rc == _lib.FQTK_ENODEV and "no CPU fallback" in msg


def test_packer_lays_nibbles_out_as_documented_and_lists_exceptions():
    """fqtk_pack_barcodes is plain host code: base k in nibble k, A 0 C 1 T 2 G 3, no-calls 7, either case; a read with
    any other byte is an exception (index + ASCII row)."""
    import numpy as np
    lib = _lib.load()
    for L in (1, 7, 8, 10, 16, 17, 20):
        ps = lib.fqtk_packed_stride(L)
        assert ps % 4 == 0 and ps >= (L + 1) // 2
        rng = np.random.default_rng(L)
        n = 500
        rows = rng.choice(np.frombuffer(b"ACGTNacgtn.", dtype=np.uint8), (n, L + 3))
        bad = rng.choice(n, 20, replace=False)
        for i in bad:
            rows[i, int(rng.integers(0, L))] = int(rng.choice(np.frombuffer(b"RYKMSWBDHVU*x\0", dtype=np.uint8)))
        packed = np.zeros((n, ps), dtype=np.uint8)
        exc_i = np.zeros(n, dtype=np.uint32)
        exc_r = np.zeros((n, L), dtype=np.uint8)
        k = C.c_uint64(0)
        assert lib.fqtk_pack_barcodes(rows.ctypes.data, L + 3, L, n, packed.ctypes.data, ps, exc_i.ctypes.data, exc_r.ctypes.data, n, C.byref(k)) == 0
        assert sorted(exc_i[:k.value].tolist()) == sorted(bad.tolist())
        code = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3, ord("N"): 7, ord("."): 7}
        for i in range(n):
            if i in set(bad.tolist()):
                j = exc_i[:k.value].tolist().index(i)
                assert bytes(exc_r[j]) == bytes(rows[i, :L])
                continue
            for b in range(L):
                nib = (int(packed[i, b // 2]) >> (4 * (b & 1))) & 0xF
                assert nib == code[int(rows[i, b]) & 0xDF if rows[i, b] != ord(".") else ord(".")], (L, i, b)
        # no room for the exceptions is an error, not an overflow
        assert lib.fqtk_pack_barcodes(rows.ctypes.data, L + 3, L, n, packed.ctypes.data, ps, exc_i.ctypes.data, exc_r.ctypes.data, 3, C.byref(k)) == _lib.FQTK_ENOMEM


def test_pack_barcodes_simd_blocks_equal_the_table_lookup_per_base():
    """fqtk_pack_barcodes is host code (no GPU needed): 4 bits per base, code = bits 1..3 of the byte for A C G T N in either
    case and '.', anything else sends the row along as an exception.  The SSSE3 block path (16 / 8 bases at a time) must
    agree with the per-base definition on every length, stride and byte value."""
    import numpy as np
    lib = _lib.load()
    rng = np.random.default_rng(7)
    alphabet = np.frombuffer(b"ACGTNacgtn.", dtype=np.uint8)
    code = np.full(256, 0xFF, dtype=np.uint8)
    for ch, c in zip(b"ACTG", range(4)):
        code[ch] = code[ch | 0x20] = c
    code[ord("N")] = code[ord("n")] = code[ord(".")] = 7
    for L in list(range(1, 34)) + [40, 64, 128]:
        for stride in (L, L + 3, (L + 3) // 4 * 4):
            n = 301                                            # (odd: the two-rows-per-step path of 16-base rows ends on a single row)
            obs = alphabet[rng.integers(0, alphabet.size, size=(n, stride))].copy()
            dirty = rng.random(n) < 0.2                       # rows with an IUPAC code / junk byte somewhere in the barcode
            for i in np.nonzero(dirty)[0]:
                obs[i, rng.integers(0, L)] = rng.choice(np.frombuffer(b"RYKMSWBDHVU#\x00\xff\xdf-", dtype=np.uint8))
            if stride > L:
                obs[:, L:] = rng.integers(0, 256, size=(n, stride - L))   # pad bytes are not part of the barcode
            ps = int(lib.fqtk_packed_stride(L))
            packed = np.full((n, ps), 0xAA, dtype=np.uint8)
            ei = np.zeros(n, dtype=np.uint32)
            er = np.zeros((n, L), dtype=np.uint8)
            k = C.c_uint64(0)
            assert lib.fqtk_pack_barcodes(obs.ctypes.data, stride, L, n, packed.ctypes.data, ps, ei.ctypes.data, er.ctypes.data, n, C.byref(k)) == 0
            c = code[obs[:, :L]]
            bad = (c == 0xFF).any(axis=1)
            want_exc = np.nonzero(bad)[0]
            assert k.value == len(want_exc) and np.array_equal(ei[:k.value], want_exc) and np.array_equal(er[:k.value], obs[want_exc, :L])
            nib = np.zeros((n, 2 * ps), dtype=np.uint8)
            nib[:, :L] = c & 7
            want = nib[:, 0::2] | (nib[:, 1::2] << 4)
            ok = ~bad                                          # (an exception row's packed bytes are not read by the device)
            assert np.array_equal(packed[ok], want[ok]), (L, stride)
