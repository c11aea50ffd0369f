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
    assert lib.fqtk_abi_version() == 3
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
