#!/usr/bin/env python3
"""Developer tool: throughput of the GPU BGZF block compressor alone (include/fqtk_bgzf.h) on FASTQ text resident
in HBM -- blocks, descriptors and outputs in device memory, so no PCIe in the timed region -- and from page-locked
host memory (the way `fqtk demux --gpu-bgzf` feeds it).  Prints one JSON object.
    python tools/bgzf_bench.py [--blocks 4096] [--reps 5]"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fqtk_amd import _lib  # noqa: E402


def fastq_text(n_records, rng, qual=b"FFFFFFFFFF:,#IIJJ<<AA"):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    q = np.frombuffer(qual, dtype=np.uint8)
    recs = []
    for i in range(n_records):
        recs.append(b"@inst:1:FC:1:%010d 1:N:0:ACGTACGT+TTGCAATG\n%s\n+\n%s\n" % (
            i, acgt[rng.integers(0, 4, 150)].tobytes(), q[rng.integers(0, len(q), 150)].tobytes()))
    return b"".join(recs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--const-qual", action="store_true", help="every quality 'I' (scope E's synthetic records: long runs)")
    ap.add_argument("--binned-qual", action="store_true", help="qualities as a NovaSeq writes them: mostly F, now and then : , # (short runs; tools/bgzf_ratio.py's second text)")
    ap.add_argument("--hbm-only", action="store_true", help="no pass with the input in pinned host memory (tools/bgzf_phases.sh: the kernel's own phases)")
    a = ap.parse_args()
    print(json.dumps(measure(a.blocks, a.reps, "binned" if a.binned_qual else a.const_qual, ("hbm",) if a.hbm_only else ("hbm", "pinned_host"))))


def measure(blocks=4096, reps=5, const_qual=False, where_list=("hbm", "pinned_host")):
    """GB/s of input and output / input of fqtk::bgzf::deflate_kernel over `blocks` 65 280-byte blocks of Illumina-style text
    (also bench.py's scopes.bgzf_kernel)."""
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(1)
    text = fastq_text(4000, rng, b"F" * 40 + b":,#") if const_qual == "binned" else (fastq_text(4000, rng, b"I") if const_qual else fastq_text(4000, rng))
    uniq = [text[o:o + 65280] for o in range(0, len(text) - 65280, 65280)]
    n = blocks
    host_in = np.zeros((n, 65536), dtype=np.uint8)
    for i in range(n):
        b = uniq[i % len(uniq)]
        host_in[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
    z = C.c_void_p()
    assert lib.fqtk_bgzf_create(0, C.byref(z)) == 0
    out = {"blocks": n, "block_bytes": 65280}
    for where in where_list:
        if where == "hbm":
            d_in = torch.from_numpy(host_in).cuda()
            d_out = torch.zeros((n, 65536), dtype=torch.uint8, device="cuda")
            d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
            p_in, p_out, p_len = d_in.data_ptr(), d_out.data_ptr(), d_len.data_ptr()
            desc_host = np.zeros((n, 3), dtype=np.uint64)
            desc_host[:, 0] = p_in + np.arange(n, dtype=np.uint64) * 65536
            desc_host[:, 1] = p_out + np.arange(n, dtype=np.uint64) * 65536
            desc_host[:, 2] = 65280
            d_desc = torch.from_numpy(desc_host.view(np.int64)).cuda()
            p_desc = d_desc.data_ptr()
        else:
            bufs = []
            for nbytes in (n * 65536, n * 65536, n * 24, n * 4):
                p = C.c_void_p()
                assert lib.fqtk_pinned_alloc(nbytes, C.byref(p)) == 0
                bufs.append(p)
            p_in, p_out, p_desc, p_len = (b.value for b in bufs)
            C.memmove(p_in, host_in.ctypes.data, host_in.nbytes)
            desc_host = np.zeros((n, 3), dtype=np.uint64)
            desc_host[:, 0] = p_in + np.arange(n, dtype=np.uint64) * 65536
            desc_host[:, 1] = p_out + np.arange(n, dtype=np.uint64) * 65536
            desc_host[:, 2] = 65280
            C.memmove(p_desc, desc_host.ctypes.data, desc_host.nbytes)
        times = []
        t_settle = time.perf_counter()   # (untimed) 60 ms of launches: the device's clocks settle (profiles/r06_bench_window.txt)
        while time.perf_counter() - t_settle < 0.06:
            assert lib.fqtk_bgzf_deflate_enqueue(z, 0, p_desc, n, p_len) == 0
            assert lib.fqtk_bgzf_wait(z, 0) == 0
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            assert lib.fqtk_bgzf_deflate_enqueue(z, 0, p_desc, n, p_len) == 0
            assert lib.fqtk_bgzf_wait(z, 0) == 0
            times.append(time.perf_counter() - t0)
        dt = min(times[1:])
        if where == "hbm":
            lens = d_len.cpu().numpy()
            payload0 = d_out[0, :int(lens[0])].cpu().numpy().tobytes()
        else:
            lens = np.ctypeslib.as_array(C.cast(p_len, C.POINTER(C.c_uint32)), (n,)).copy()
            payload0 = C.string_at(p_out, int(lens[0]))
        assert zlib.decompress(payload0, -15) == uniq[0]
        out[where] = {"seconds": round(dt, 5), "GB_per_s_in": round(n * 65280 / dt / 1e9, 2),
                      "blocks_per_s": round(n / dt), "ratio": round(float(lens.sum()) / (n * 65280), 4)}
    cpu = []
    t0 = time.perf_counter()
    for b in uniq[:20]:
        cpu.append(len(zlib.compress(b, 5)))
    out["zlib_level5_1core_GB_per_s"] = round(20 * 65280 / (time.perf_counter() - t0) / 1e9, 3)
    lib.fqtk_bgzf_destroy(z)
    return out


if __name__ == "__main__":
    main()
