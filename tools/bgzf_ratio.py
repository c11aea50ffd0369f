#!/usr/bin/env python3
"""Developer tool (CPU only): compression ratio of the BGZF compressor's algorithm -- the same phase functions the
kernel runs, through the host emulation -- on two kinds of FASTQ text, next to zlib levels 1 / 5 / 6.  Optional
extra -D flags build a variant of the algorithm (e.g. -DFQTK_BGZF_DROP=8 drops candidate 3).
    python tools/bgzf_ratio.py [-DFLAG ...]"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fastq_text(n_records, rng, qual):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    q = np.frombuffer(qual, dtype=np.uint8)
    recs = []
    x = 1000
    for i in range(n_records):   # as a sequencer writes them: tile by tile, x ascending within a tile, y anywhere
        qs = q[rng.integers(0, len(q), 150)].tobytes()
        x += int(rng.integers(0, 12))
        if x > 30000:
            x = 1000
        recs.append(b"@A00123:45:HXXXXXXXX:1:%04d:%d:%d 1:N:0:ACGTACGT+TTGCAATG\n%s\n+\n%s\n" % (
            1101 + i // 2500, x, rng.integers(1000, 30000), acgt[rng.integers(0, 4, 150)].tobytes(), qs))
    return b"".join(recs)


def main():
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    so = os.path.join(tempfile.mkdtemp(), "libshim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", *flags, "-o", so,
                           os.path.join(ROOT, "fqtk_amd/csrc/host/host_capi.cpp"), "-lz", "-ldl"])
    lib = C.CDLL(so)
    fn = lib.fqtk_host_bgzf_deflate_level
    fn.restype = C.c_int64
    rng = np.random.default_rng(1)
    sets = {"varied qualities (21 symbols, uniform)": fastq_text(3000, rng, b"FFFFFFFFFF:,#IIJJ<<AA"),
            "binned qualities (mostly F)": fastq_text(3000, rng, b"F" * 40 + b":,#")}
    for name, text in sets.items():
        blocks = [text[o:o + 65280] for o in range(0, len(text) - 65280, 65280)][:12]
        out = (C.c_uint8 * 70000)()
        stored = C.c_int(0)
        tot_in = sum(len(b) for b in blocks)
        ours = {}
        for level in (1, 5):   # parse effort 0 / 1
            tot_out = 0
            for b in blocks:
                n = fn(b, C.c_uint32(len(b)), out, C.c_size_t(70000), C.byref(stored), 0, level)
                assert n > 0 and zlib.decompress(bytes(out[:n]), -15) == b
                tot_out += n
            ours[level] = tot_out / tot_in
        z = {lvl: sum(len(zlib.compress(b, lvl)) for b in blocks) / tot_in for lvl in (1, 5, 6)}
        print(f"{name}: this at levels 1-3 {ours[1]:.4f}, 4+ {ours[5]:.4f} | zlib-1 {z[1]:.4f} zlib-5 {z[5]:.4f} zlib-6 {z[6]:.4f}")


if __name__ == "__main__":
    main()
