"""ctypes front-end of oracle/ref_literal.c plus an independent numpy restatement (`ref_simple`).

TEST INFRASTRUCTURE ONLY (see ref_literal.c header).  `RefLiteral` follows the reference line by line
(/root/reference/src/lib/barcode_matching.rs:55-186, bitenc.rs:432-459, mod.rs:26-92); `ref_simple_*`
is the distilled spec: mm[s] = #{i : enc(read[i]) & ~E[s][i] != 0},
best = min, best_idx = lowest index attaining it, next = second smallest with multiplicity.
The two are property-tested against each other in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "ref_literal.c")

NONE_IDX = 0xFFFF


def build(native: bool = False, force: bool = False) -> str:
    """Compile ref_literal.c with gcc.  native=True adds -march=native (used by bench.py's
    cpu_baseline leg, which rebuilds on the box it runs on)."""
    out = os.path.join(_HERE, f"liboracle_native_{_cpu_tag()}.so" if native else "liboracle.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(_SRC):
        return out
    tmp = f"{out}.{os.getpid()}.tmp"      # several processes may build at once: compile aside, rename into place
    cmd = ["gcc", "-O3", "-fPIC", "-std=c11", "-shared", "-o", tmp, _SRC]
    if native:
        cmd.insert(2, "-march=native")
    subprocess.run(cmd, check=True)
    os.replace(tmp, out)
    return out


def _cpu_tag() -> str:
    """-march=native code is only valid on the CPU model it was built on (this tree travels between boxes)."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as fh:
            lines = [ln for ln in fh if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        lines = []
    return hashlib.sha1("".join(lines).encode()).hexdigest()[:10]


def _load(native: bool = False) -> C.CDLL:
    lib = C.CDLL(build(native=native))
    u8p = C.POINTER(C.c_uint8)
    lib.oracle_is_valid_iupac.argtypes = [C.c_uint8]
    lib.oracle_is_valid_iupac.restype = C.c_int
    lib.oracle_byte_is_nocall.argtypes = [C.c_uint8]
    lib.oracle_byte_is_nocall.restype = C.c_int
    lib.oracle_encode.argtypes = [C.c_char_p, C.c_size_t, u8p, C.POINTER(C.c_uint32)]
    lib.oracle_encode.restype = C.c_size_t
    lib.oracle_hamming_vals.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_uint32]
    lib.oracle_hamming_vals.restype = C.c_int64
    lib.oracle_count_mismatches.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint8]
    lib.oracle_count_mismatches.restype = C.c_int
    lib.oracle_matcher_new.argtypes = [C.POINTER(C.c_char_p), C.c_size_t, C.c_uint8, C.c_uint8, C.c_int,
                                       C.c_char_p, C.c_size_t]
    lib.oracle_matcher_new.restype = C.c_void_p
    lib.oracle_matcher_free.argtypes = [C.c_void_p]
    lib.oracle_matcher_free.restype = None
    lib.oracle_matcher_max_ns.argtypes = [C.c_void_p]
    lib.oracle_matcher_max_ns.restype = C.c_size_t
    lib.oracle_matcher_cache_hits.argtypes = [C.c_void_p]
    lib.oracle_matcher_cache_hits.restype = C.c_uint64
    lib.oracle_matcher_cache_misses.argtypes = [C.c_void_p]
    lib.oracle_matcher_cache_misses.restype = C.c_uint64
    lib.oracle_assign.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), u8p, u8p]
    lib.oracle_assign.restype = C.c_int
    lib.oracle_assign_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_uint64)]
    lib.oracle_assign_batch.restype = C.c_int
    return lib


_LIBS = {}


def lib(native: bool = False) -> C.CDLL:
    if native not in _LIBS:
        _LIBS[native] = _load(native)
    return _LIBS[native]


class OracleLengthError(Exception):
    """The reference panics here: observed barcode length != expected barcode length
    (barcode_matching.rs:95-107)."""


def encode(bases: bytes) -> Tuple[list, list]:
    n = len(bases)
    vals = (C.c_uint8 * max(n, 1))()
    blocks = (C.c_uint32 * max((n + 7) // 8, 1))()
    nb = lib().oracle_encode(bases, n, vals, blocks)
    return list(vals[:n]), list(blocks[:nb])


def hamming_vals(a: Sequence[int], b: Sequence[int], max_mismatches: int) -> int:
    aa = (C.c_uint8 * max(len(a), 1))(*a)
    bb = (C.c_uint8 * max(len(b), 1))(*b)
    return int(lib().oracle_hamming_vals(aa, len(a), bb, len(b), max_mismatches))


def count_mismatches(observed: bytes, expected: bytes, max_mismatches: int = 255) -> int:
    r = lib().oracle_count_mismatches(observed, len(observed), expected, len(expected), max_mismatches)
    if r < 0:
        raise OracleLengthError(
            f"Read barcode length ({len(observed)}) differs from expected barcode length ({len(expected)})")
    return r


def is_valid_iupac(b: int) -> bool:
    return bool(lib().oracle_is_valid_iupac(b))


class RefLiteral:
    """Literal restatement of BarcodeMatcher (barcode_matching.rs:29-186)."""

    def __init__(self, barcodes: Sequence[str], max_mismatches: int, min_mismatch_delta: int,
                 use_cache: bool = True, native: bool = False):
        self._lib = lib(native)
        arr = (C.c_char_p * max(len(barcodes), 1))(*[b.encode() for b in barcodes])
        err = C.create_string_buffer(256)
        self._h = self._lib.oracle_matcher_new(arr, len(barcodes), max_mismatches, min_mismatch_delta,
                                               1 if use_cache else 0, err, 256)
        if not self._h:
            raise ValueError(err.value.decode())
        self.n_samples = len(barcodes)
        self.barcode_len = len(barcodes[0])

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.oracle_matcher_free(h)
            self._h = None

    @property
    def max_ns_in_barcodes(self) -> int:
        return int(self._lib.oracle_matcher_max_ns(self._h))

    @property
    def cache_stats(self) -> Tuple[int, int]:
        return (int(self._lib.oracle_matcher_cache_hits(self._h)),
                int(self._lib.oracle_matcher_cache_misses(self._h)))

    def assign(self, read: bytes) -> Optional[Tuple[int, int, int]]:
        bi = C.c_uint64(0)
        b = C.c_uint8(0)
        nx = C.c_uint8(0)
        rc = self._lib.oracle_assign(self._h, read, len(read), C.byref(bi), C.byref(b), C.byref(nx))
        if rc < 0:
            raise OracleLengthError("observed barcode longer than expected barcode")
        if rc == 0:
            return None
        return (int(bi.value), int(b.value), int(nx.value))

    def assign_batch(self, obs: np.ndarray, lens: Optional[np.ndarray] = None):
        """obs: uint8 [n, stride] C-contiguous.  Returns (idx u16, best u8, next u8, counts u64[S+1]).
        None rows are (0xFFFF, 255, 255)."""
        obs = np.ascontiguousarray(obs, dtype=np.uint8)
        n, stride = obs.shape
        idx = np.empty(n, dtype=np.uint16)
        best = np.empty(n, dtype=np.uint8)
        nxt = np.empty(n, dtype=np.uint8)
        counts = np.zeros(self.n_samples + 1, dtype=np.uint64)
        lp = None
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint32)
            lp = lens.ctypes.data
        err_index = C.c_uint64(0)
        rc = self._lib.oracle_assign_batch(self._h, obs.ctypes.data, stride, lp, n, idx.ctypes.data,
                                           best.ctypes.data, nxt.ctypes.data, counts.ctypes.data,
                                           C.byref(err_index))
        if rc < 0:
            raise OracleLengthError(f"read {err_index.value}: observed barcode longer than expected")
        return idx, best, nxt, counts


# ------------------------------------------------------------------------------------------------
# ref_simple: independent numpy restatement of the distilled spec (SURVEY.md 8a "normative spec").
# ------------------------------------------------------------------------------------------------

def _enc_table() -> np.ndarray:
    """enc(b) for all 256 byte values: N/n/. -> 15; else IUPAC mask of the upper-cased byte; else 0
    (mod.rs:26-61)."""
    t = np.zeros(256, dtype=np.uint8)
    masks = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "M": 3, "R": 5, "W": 9, "S": 6, "Y": 10, "K": 12,
             "V": 7, "H": 11, "D": 13, "B": 14, "N": 15}
    for ch, v in masks.items():
        t[ord(ch)] = v
        t[ord(ch.lower())] = v
    t[ord(".")] = 15
    return t


ENC = _enc_table()


def ref_simple_assign_batch(barcodes: Sequence[str], max_mismatches: int, min_mismatch_delta: int,
                            obs: np.ndarray, lens: Optional[np.ndarray] = None, chunk: int = 1 << 16):
    """Vectorised spec.  obs uint8 [n, stride>=L]; rows use their first L bytes when len >= L.
    Rows with len < L are None; rows with len > L raise unless the no-call prefilter rejects them
    (barcode_matching.rs:165-172).  Returns (idx, best, next, counts) with None = (0xFFFF,255,255)."""
    S = len(barcodes)
    L = len(barcodes[0])
    E = np.stack([ENC[np.frombuffer(b.upper().encode(), dtype=np.uint8)] for b in barcodes])  # [S, L]
    notE = (~E) & 0xF
    max_ns = max(sum(1 for ch in b.upper() if ch in "N.") for b in barcodes)
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    n, stride = obs.shape
    idx = np.full(n, NONE_IDX, dtype=np.uint16)
    best = np.full(n, 255, dtype=np.uint8)
    nxt = np.full(n, 255, dtype=np.uint8)
    if lens is None:
        lens = np.full(n, stride, dtype=np.uint32)
    lens = np.asarray(lens)
    pos = np.arange(stride)[None, :]
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        o_raw = obs[lo:hi]
        ln = lens[lo:hi]
        valid = pos < ln[:, None]
        nocall = (((o_raw == ord("N")) | (o_raw == ord("n")) | (o_raw == ord("."))) & valid).sum(axis=1)
        short = ln < L
        pre = nocall > (max_mismatches + max_ns)
        longer = (ln > L) & ~pre
        if longer.any():
            raise OracleLengthError(f"read {lo + int(np.argmax(longer))}: observed longer than expected")
        live = ~(short | pre)
        o = ENC[o_raw[:, :L]] if stride >= L else np.zeros((hi - lo, L), dtype=np.uint8)
        mm = ((o[:, None, :] & notE[None, :, :]) != 0).sum(axis=2)  # [m, S]
        mm = np.minimum(mm, 255)
        bi = mm.argmin(axis=1)  # first index attaining the min
        b = mm[np.arange(hi - lo), bi]
        if S > 1:
            mm2 = mm.copy()
            mm2[np.arange(hi - lo), bi] = 1 << 20
            nx = np.minimum(mm2.min(axis=1), 255)
        else:
            nx = np.full(hi - lo, 255)
        ok = live & (b <= max_mismatches) & ((nx - b) >= min_mismatch_delta)
        idx[lo:hi][ok] = bi[ok].astype(np.uint16)
        best[lo:hi][ok] = b[ok].astype(np.uint8)
        nxt[lo:hi][ok] = nx[ok].astype(np.uint8)
    counts = np.zeros(S + 1, dtype=np.uint64)
    sel = np.where(idx == NONE_IDX, S, idx.astype(np.int64))
    np.add.at(counts, sel, 1)
    return idx, best, nxt, counts
