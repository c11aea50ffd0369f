#!/usr/bin/env python3
"""Developer tool: throughput of the GPU BGZF member decoder alone (include/fqtk_inflate.h) on members resident in HBM
(payloads, descriptors and text in device memory: no PCIe in the timed region).  Prints one JSON object.
    python tools/inflate_bench.py [--members 8192] [--reps 5] [--level 6]"""
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
from tools.bgzf_bench import fastq_text  # noqa: E402


def measure(members=8192, reps=5, level=6, const_qual=False, check=True):
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(1)
    text = fastq_text(8000, rng, b"I") if const_qual else fastq_text(8000, rng)
    uniq = [text[o:o + 65280] for o in range(0, len(text) - 65280, 65280)]
    comp = []
    for t in uniq:
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp.append(c.compress(t) + c.flush())
    file_bytes = bytearray()
    desc = np.zeros(members, dtype=[("payload_off", "<u8"), ("out_off", "<u8"), ("payload_len", "<u4"), ("isize", "<u4"), ("crc", "<u4"), ("reserved", "<u4")])
    out_off = 0
    for i in range(members):
        t, p = uniq[i % len(uniq)], comp[i % len(uniq)]
        file_bytes += b"\0" * 18
        desc[i] = (len(file_bytes), out_off, len(p), len(t), zlib.crc32(t), 0)
        file_bytes += p + b"\0" * 8
        out_off += len(t)
    file_bytes += b"\0" * 8
    d_in = torch.from_numpy(np.frombuffer(bytes(file_bytes), dtype=np.uint8).copy()).cuda()
    d_desc = torch.from_numpy(desc.view(np.uint8).copy()).cuda()
    d_out = torch.zeros(out_off + 64, dtype=torch.uint8, device="cuda")
    d_stat = torch.zeros(members, dtype=torch.int32, device="cuda")
    d_lines = torch.zeros(members, dtype=torch.int32, device="cuda")
    z = C.c_void_p()
    assert lib.fqtk_inflate_create(0, C.byref(z)) == 0, lib.fqtk_inflate_last_error()
    best = None
    t_settle = time.perf_counter()   # (untimed) 60 ms of launches: the device's clocks settle (profiles/r06_bench_window.txt)
    while time.perf_counter() - t_settle < 0.06:
        assert lib.fqtk_inflate_enqueue(z, 0, d_in.data_ptr(), len(file_bytes), d_desc.data_ptr(), members, d_out.data_ptr(), d_stat.data_ptr(), d_lines.data_ptr()) == 0
        assert lib.fqtk_inflate_wait(z, 0) == 0
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert lib.fqtk_inflate_enqueue(z, 0, d_in.data_ptr(), len(file_bytes), d_desc.data_ptr(), members, d_out.data_ptr(), d_stat.data_ptr(), d_lines.data_ptr()) == 0
        assert lib.fqtk_inflate_wait(z, 0) == 0
        dt = time.perf_counter() - t0
        if r:
            best = dt if best is None else min(best, dt)
    if check:
        assert int(d_stat.abs().sum().item()) == 0, "a member failed"
        got = bytes(d_out[:len(uniq[0])].cpu().numpy())
        assert got == uniq[0]
        assert int(d_lines.sum().item()) == sum(uniq[i % len(uniq)].count(b"\n") for i in range(members))
    lib.fqtk_inflate_destroy(z)
    return {"members": members, "level": level, "text_bytes": out_off, "compressed_bytes": len(file_bytes), "ratio": round(len(file_bytes) / out_off, 4),
            "best_ms": round(best * 1e3, 3), "text_GBps": round(out_off / best / 1e9, 2), "compressed_GBps": round(len(file_bytes) / best / 1e9, 2)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--const-qual", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="ablation builds (tools/ab_inflate.sh) decode wrong bytes")
    a = ap.parse_args()
    print(json.dumps(measure(a.members, a.reps, a.level, a.const_qual, not a.no_check)))
