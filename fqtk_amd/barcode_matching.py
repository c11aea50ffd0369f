"""BarcodeMatcher / BarcodeMatch: Python mirror of the reference's matcher API over the C ABI.

Mirrors /root/reference/src/lib/barcode_matching.rs: `BarcodeMatcher::new(samples, max_mismatches,
min_mismatch_delta, use_cache)` (:55-86) and `assign(read_bases) -> Option<BarcodeMatch>` (:165-186),
so parity tests read like the reference's own tests (:326-447).  Every call lands in the hand-written
HIP kernels behind include/fqtk_match.h; `use_cache` selects the precomputed complete memo (True, demux's
setting) or the exhaustive scan for every read (False); results are identical, as in the reference
(:174-181).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Union

import numpy as np

from . import _lib
from .samples import Sample

NO_MATCH = _lib.FQTK_NO_MATCH
MATCH_DTYPE = np.dtype([("idx", "<u2"), ("best", "u1"), ("next", "u1")])


class FqtkError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"fqtk error {code}: {message}")
        self.code = code
        self.message = message


class FqtkLengthError(FqtkError):
    """The reference panics here: "Read barcode (..) length (n) differs from expected barcode (..)
    length (m) for sample .." (barcode_matching.rs:95-107)."""


def _check(rc: int) -> None:
    if rc == _lib.FQTK_OK:
        return
    msg = _lib.last_error()
    if rc == _lib.FQTK_ELEN:
        raise FqtkLengthError(rc, msg)
    if rc == _lib.FQTK_EINVAL:
        raise ValueError(msg)
    raise FqtkError(rc, msg)


@dataclass(frozen=True)
class BarcodeMatch:
    """barcode_matching.rs:16-25."""
    best_match: int
    best_mismatches: int
    next_best_mismatches: int


class BarcodeMatcher:
    def __init__(self, samples: Sequence[Union[Sample, str]], max_mismatches: int,
                 min_mismatch_delta: int, use_cache: bool = True, device: int = 0):
        if not (0 <= max_mismatches <= 255 and 0 <= min_mismatch_delta <= 255):
            raise ValueError("max_mismatches / min_mismatch_delta must fit in u8")  # demux.rs:923-924
        self._h = None
        self._lib = _lib.load()
        barcodes = [s.barcode if isinstance(s, Sample) else s for s in samples]
        self.barcodes = list(barcodes)
        n = len(barcodes)
        arr = (C.c_char_p * max(n, 1))(*[b.encode("latin-1") for b in barcodes])
        L = len(barcodes[0].encode("latin-1")) if n else 0
        h = C.c_void_p()
        _check(self._lib.fqtk_matcher_create(arr, n, L, max_mismatches, min_mismatch_delta, device,
                                             C.byref(h)))
        self._h = h
        if n and all(isinstance(s, Sample) for s in samples):   # ids word the length-error message (:95-107)
            ids = (C.c_char_p * n)(*[s.sample_id.encode("latin-1") for s in samples])
            _check(self._lib.fqtk_matcher_set_sample_ids(h, ids))
        _check(self._lib.fqtk_matcher_set_use_cache(h, 1 if use_cache else 0))
        self.use_cache = use_cache
        self.max_mismatches = max_mismatches
        self.min_mismatch_delta = min_mismatch_delta
        self.device = device

    # ---- lifetime --------------------------------------------------------------------------------
    def close(self) -> None:
        if self._h:
            self._lib.fqtk_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- introspection ---------------------------------------------------------------------------
    @property
    def n_samples(self) -> int:
        return int(self._lib.fqtk_matcher_n_samples(self._h))

    @property
    def barcode_len(self) -> int:
        return int(self._lib.fqtk_matcher_barcode_len(self._h))

    @property
    def max_ns_in_barcodes(self) -> int:
        return int(self._lib.fqtk_matcher_max_ns_in_barcodes(self._h))

    @property
    def memo_entries(self) -> int:
        """Entries of the precomputed complete memo (0 = exhaustive scan only)."""
        return int(self._lib.fqtk_matcher_memo_entries(self._h))

    @property
    def memo_candidates(self) -> int:
        """Strings enumerated and scanned on the device to build the memo (0 when no memo was built)."""
        return int(self._lib.fqtk_matcher_memo_candidates(self._h))

    MEMO_NONE, MEMO_TABLE, MEMO_LDS = 0, 1, 2

    @property
    def memo_kind(self) -> int:
        """Form of the memo the next batch uses: MEMO_NONE (scan), MEMO_TABLE (HBM/L2), MEMO_LDS (LDS-resident)."""
        return int(self._lib.fqtk_matcher_memo_kind(self._h))

    @memo_kind.setter
    def memo_kind(self, kind: int) -> None:
        _check(self._lib.fqtk_matcher_set_memo_kind(self._h, int(kind)))

    @property
    def memo_direct_bytes(self) -> int:
        """2 or 4 when MEMO_TABLE is the direct-indexed variant (barcodes of <= 10 bases), else 0."""
        return int(self._lib.fqtk_matcher_memo_direct_bytes(self._h))

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    # ---- the reference's call shape: one read in, Option<BarcodeMatch> out ------------------------
    def assign(self, read_bases: bytes) -> Optional[BarcodeMatch]:
        out = _lib.fqtk_match_t()
        _check(self._lib.fqtk_matcher_assign1(self._h, bytes(read_bases), len(read_bases), C.byref(out)))
        if out.idx == NO_MATCH:
            return None
        return BarcodeMatch(int(out.idx), int(out.best), int(out.next))

    # ---- batch forms -----------------------------------------------------------------------------
    def assign_batch(self, obs: np.ndarray, lens: Optional[np.ndarray] = None, counts: bool = True):
        """obs: uint8 [n, stride] (host).  Returns (matches[MATCH_DTYPE n], counts u64[S+1] or None)."""
        obs = np.ascontiguousarray(obs, dtype=np.uint8)
        assert obs.ndim == 2
        n, stride = obs.shape
        out = np.empty(n, dtype=MATCH_DTYPE)
        cnt = np.zeros(self.n_samples + 1, dtype=np.uint64) if counts else None
        lp = None
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.uint32)
            assert lens.shape == (n,)
            lp = lens.ctypes.data
        _check(self._lib.fqtk_matcher_assign_batch(self._h, obs.ctypes.data, stride, lp, n,
                                                   out.ctypes.data, cnt.ctypes.data if counts else None))
        return out, cnt

    # ---- packed input: 4 bits per base over PCIe (fqtk_pack_barcodes / fqtk_matcher_enqueue_packed) ----------
    def pack(self, obs: np.ndarray):
        """obs: uint8 [n, stride >= barcode_len] ASCII rows -> (packed uint8 [n, packed_stride], exc_index u32[k], exc_rows u8[k, L])."""
        obs = np.ascontiguousarray(obs, dtype=np.uint8)
        n, stride = obs.shape
        L = self.barcode_len
        ps = int(self._lib.fqtk_packed_stride(L))
        packed = np.empty((n, ps), dtype=np.uint8)
        exc_index = np.empty(max(n, 1), dtype=np.uint32)
        exc_rows = np.empty((max(n, 1), L), dtype=np.uint8)
        n_exc = C.c_uint64(0)
        _check(self._lib.fqtk_pack_barcodes(obs.ctypes.data, stride, L, n, packed.ctypes.data, ps, exc_index.ctypes.data,
                                            exc_rows.ctypes.data, n, C.byref(n_exc)))
        k = int(n_exc.value)
        return packed, exc_index[:k].copy(), exc_rows[:k].copy()

    def assign_batch_packed(self, packed: np.ndarray, exc_index: np.ndarray, exc_rows: np.ndarray, slot: int = 0):
        """The packed entry, synchronously: returns (matches, counts of this call)."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        exc_index = np.ascontiguousarray(exc_index, dtype=np.uint32)
        exc_rows = np.ascontiguousarray(exc_rows, dtype=np.uint8)
        n, ps = packed.shape
        out = np.empty(n, dtype=MATCH_DTYPE)
        scratch = np.zeros(self.n_samples + 1, dtype=np.uint64)
        _check(self._lib.fqtk_matcher_counts(self._h, scratch.ctypes.data))          # start from an empty accumulator
        _check(self._lib.fqtk_matcher_enqueue_packed(self._h, slot, packed.ctypes.data, ps, n,
                                                     exc_index.ctypes.data if exc_index.size else None,
                                                     exc_rows.ctypes.data if exc_index.size else None, exc_index.size, out.ctypes.data))
        _check(self._lib.fqtk_matcher_wait(self._h, slot))
        cnt = np.zeros(self.n_samples + 1, dtype=np.uint64)
        _check(self._lib.fqtk_matcher_counts(self._h, cnt.ctypes.data))
        return out, cnt

    def assign_batch_device(self, d_obs: int, stride: int, n: int, d_out: int, d_counts: int = 0,
                            d_lens: int = 0, stream: int = 0) -> None:
        """Zero-copy form: raw device pointers (e.g. torch tensor .data_ptr()) and a hipStream_t
        handle (e.g. torch.cuda.current_stream().cuda_stream).  Asynchronous."""
        _check(self._lib.fqtk_matcher_assign_batch_device(self._h, d_obs, stride, d_lens or None, n,
                                                          d_out, d_counts or None, stream or None))

    def poll_error(self, stream: int = 0) -> None:
        idx = C.c_uint64(0)
        _check(self._lib.fqtk_matcher_poll_error(self._h, stream or None, C.byref(idx)))
