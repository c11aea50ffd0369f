"""Synthetic workloads for the five BASELINE.json configs (bench / test harness).

Sample tables are built here (numpy, seeded); observed barcodes come from the counter-based generator
in csrc/synth.hip, which yields the SAME bytes on the device and on the host for a given
(seed, read index).  Not part of the matcher ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libfqtk_synth.so")


@dataclass(frozen=True)
class Config:
    name: str
    n_reads: int
    n_samples: int
    barcode_len: int
    max_mismatches: int
    min_mismatch_delta: int
    iupac: bool = False
    read_structures: Tuple[str, ...] = ()
    stride: int = 0      # SoA stride in bytes (barcode_len rounded up to 4)
    seed: int = 0

    @property
    def bytes_per_read(self) -> int:
        """Algorithmic HBM bytes per read of the fused ASCII-in kernel: Lpad in + 4 out."""
        return self.stride + 4


def _cfg(i, name, n, S, L, mm, d, iupac=False, rs=()):
    return Config(name, n, S, L, mm, d, iupac, tuple(rs), (L + 3) // 4 * 4, 0xF07C + i)


# BASELINE.json `configs` (index = position in that list, 1-based ids in SURVEY.md section 8)
CONFIGS = {
    1: _cfg(1, "cfg1 single-end 1M x 150bp, 8B inline, 16 samples", 1_000_000, 16, 8, 1, 2,
            rs=("8B142T",)),
    2: _cfg(2, "cfg2 paired-end 100M + I1 8bp, 96 single-index samples, mm=1", 100_000_000, 96, 8, 1, 2,
            rs=("150T", "150T", "8B")),
    3: _cfg(3, "cfg3 dual-index 400M reads, 384 samples (8+8bp), delta=2", 400_000_000, 384, 16, 1, 2,
            rs=("150T", "8B", "8B", "150T")),
    4: _cfg(4, "cfg4 10x-style 16C8B, 24 samples, 200M reads", 200_000_000, 24, 8, 1, 2,
            rs=("16C8B126T", "150T")),
    5: _cfg(5, "cfg5 IUPAC-degenerate 1536 samples x 10bp, 50M reads", 50_000_000, 1536, 10, 1, 2,
            iupac=True, rs=("10B+T",)),
}

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_DEGENERATE = np.frombuffer(b"MRWSYKVHDBN", dtype=np.uint8)


def make_barcodes(cfg: Config) -> List[str]:
    """cfg 1-4: uniform ACGT, rejection-sampled to pairwise Hamming distance >= 3.
    cfg 5: 10-mers with 1-3 positions replaced by a random degenerate IUPAC code, unique strings."""
    rng = np.random.default_rng(cfg.seed)
    S, L = cfg.n_samples, cfg.barcode_len
    chosen = np.empty((0, L), dtype=np.uint8)
    seen = set()
    out: List[np.ndarray] = []
    while len(out) < S:
        cand = _ACGT[rng.integers(0, 4, size=L)]
        if cfg.iupac:
            k = int(rng.integers(1, 4))
            pos = rng.choice(L, size=k, replace=False)
            cand = cand.copy()
            cand[pos] = _DEGENERATE[rng.integers(0, len(_DEGENERATE), size=k)]
            key = cand.tobytes()
            if key in seen:
                continue
            seen.add(key)
            out.append(cand)
            continue
        if len(out) and int(((chosen != cand[None, :]).sum(axis=1)).min()) < 3:
            continue
        out.append(cand)
        chosen = np.vstack([chosen, cand[None, :]])
    return [b.tobytes().decode() for b in out]


def zipf_cdf(S: int, s: float = 0.5) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, S + 1, dtype=np.float64), s)
    c = np.cumsum(w) / w.sum()
    cdf = np.minimum(np.floor(c * 2.0**32), 2.0**32 - 1).astype(np.uint64).astype(np.uint32)
    cdf[-1] = 0xFFFFFFFF
    return cdf


def thresholds(cfg: Config) -> np.ndarray:
    p_sample, p_n, p_sub = 0.90, 0.005, 0.01
    p_lower, p_dot = (0.001, 0.0001) if cfg.iupac else (0.0, 0.0)
    if os.environ.get("FQTK_SYNTH_PSAMPLE"):  # developer knob: fraction of reads drawn from a sample
        p_sample = float(os.environ["FQTK_SYNTH_PSAMPLE"])
    if os.environ.get("FQTK_SYNTH_PDOT"):     # developer knob: rate of '.' no-calls per base
        p_dot = float(os.environ["FQTK_SYNTH_PDOT"])
    p_iupac = float(os.environ.get("FQTK_SYNTH_PIUPAC", "0"))   # developer knob: rate of IUPAC 'R' bytes per base of a READ
    t = [p_sample, p_n, p_sub, p_lower, p_dot, p_iupac]
    return np.array([min(int(p * 2**32), 2**32 - 1) for p in t], dtype=np.uint32)


_lib = None


def _load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} missing: run `python -m fqtk_amd.build`")
        from ._lib import preload_hip_runtime
        preload_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        common = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p,
                  C.c_uint64, C.c_uint64, C.c_void_p]
        lib.fqtk_synth_fill_host.argtypes = common
        lib.fqtk_synth_fill_host.restype = C.c_int
        lib.fqtk_synth_fill_device.argtypes = common + [C.c_void_p]
        lib.fqtk_synth_fill_device.restype = C.c_int
        _lib = lib
    return _lib


class Workload:
    """A config + its sample table; generates observed barcodes for any read range."""

    def __init__(self, cfg: Config, seed_offset: int = 0):
        self.cfg = cfg
        self.barcodes = make_barcodes(cfg)
        self._bc = np.frombuffer("".join(self.barcodes).encode(), dtype=np.uint8).copy()
        self._cdf = zipf_cdf(cfg.n_samples)
        self._thr = thresholds(cfg)
        self.seed = (cfg.seed * 0x9E3779B97F4A7C15 + seed_offset * 0xD6E8FEB86659FD93) & (2**64 - 1)

    def fill_host(self, start: int, n: int) -> np.ndarray:
        out = np.empty((n, self.cfg.stride), dtype=np.uint8)
        rc = _load().fqtk_synth_fill_host(self._bc.ctypes.data, self._cdf.ctypes.data, self.cfg.n_samples,
                                          self.cfg.barcode_len, self.cfg.stride, self.seed,
                                          self._thr.ctypes.data, start, n, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"fqtk_synth_fill_host failed: {rc}")
        return out

    def fill_device(self, start: int, n: int, d_out: int, stream: int = 0) -> None:
        rc = _load().fqtk_synth_fill_device(self._bc.ctypes.data, self._cdf.ctypes.data,
                                            self.cfg.n_samples, self.cfg.barcode_len, self.cfg.stride,
                                            self.seed, self._thr.ctypes.data, start, n, d_out,
                                            stream or None)
        if rc != 0:
            raise RuntimeError(f"fqtk_synth_fill_device failed: {rc}")
