#!/usr/bin/env python3
"""Scopes B and E of SURVEY.md 8(d) (the headline bench.py is scope K, HBM-resident):
  B  boundary level: pinned host SoA barcodes -> fqtk_matcher_enqueue/wait -> host results (PCIe incl.)
  E  end to end: `fqtk demux` on synthetic dual-index FASTQ files (gz or plain) -> per-sample BGZF
Run on the GPU box:  python tools/scope_bench.py [--templates 2000000] [--threads 32] [--gz]
Prints one JSON object; numbers go to DESIGN.md, never into bench.py's `value`."""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fqtk_amd import BarcodeMatcher, _lib, synth  # noqa: E402


def scope_b(cfg_id=3, n_chunk=8_000_000, chunks=12):
    cfg = synth.CONFIGS[cfg_id]
    w = synth.Workload(cfg)
    lib = _lib.load()
    m = BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    bufs = []
    for s in range(2):
        po, pr = C.c_void_p(), C.c_void_p()
        assert lib.fqtk_pinned_alloc(n_chunk * cfg.stride, C.byref(po)) == 0
        assert lib.fqtk_pinned_alloc(n_chunk * 4, C.byref(pr)) == 0
        host = w.fill_host(s * n_chunk, n_chunk)
        C.memmove(po, host.ctypes.data, host.nbytes)
        bufs.append((po, pr))
    for s in range(2):   # warm-up
        assert lib.fqtk_matcher_enqueue(m.handle, s, bufs[s][0], cfg.stride, None, n_chunk, bufs[s][1]) == 0
    for s in range(2):
        assert lib.fqtk_matcher_wait(m.handle, s) == 0
    t0 = time.perf_counter()
    for c in range(chunks):
        s = c % 2
        if c >= 2:
            assert lib.fqtk_matcher_wait(m.handle, s) == 0
        assert lib.fqtk_matcher_enqueue(m.handle, s, bufs[s][0], cfg.stride, None, n_chunk, bufs[s][1]) == 0
    for s in range(2):
        assert lib.fqtk_matcher_wait(m.handle, s) == 0
    dt = time.perf_counter() - t0
    reads = n_chunk * chunks
    return {"reads": reads, "seconds": round(dt, 4), "M_reads_per_s": round(reads / dt / 1e6, 1),
            "GB_per_s_over_pcie": round(reads * (cfg.stride + 4) / dt / 1e9, 2), "chunk_reads": n_chunk}


def fixed_fastq(path, n, seqs, read_no, gz):
    """n records with fixed-width fields, built as one numpy byte matrix."""
    L = seqs.shape[1]
    head = np.frombuffer(b"@inst:1:FC:1:0000000000 %d:N:0:0\n" % read_no, dtype=np.uint8)
    rec = np.empty((n, len(head) + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, :len(head)] = head
    digits = np.arange(n)[:, None] // (10 ** np.arange(9, -1, -1))[None, :] % 10
    rec[:, 13:23] = digits.astype(np.uint8) + ord("0")
    o = len(head)
    rec[:, o:o + L] = seqs
    rec[:, o + L] = ord("\n")
    rec[:, o + L + 1] = ord("+")
    rec[:, o + L + 2] = ord("\n")
    rec[:, o + L + 3:o + 2 * L + 3] = ord("I")
    rec[:, -1] = ord("\n")
    with open(path, "wb") as fh:
        fh.write(rec.tobytes())
    if gz:
        subprocess.run(["gzip", "-1", "-f", path], check=True)
        return path + ".gz"
    return path


def scope_e(n, threads, gz, tmp):
    cfg = synth.CONFIGS[3]
    w = synth.Workload(cfg)
    bcs = w.fill_host(0, n)
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    t = acgt[rng.integers(0, 4, size=(n, 150))]
    files = [fixed_fastq(os.path.join(tmp, "R1.fastq"), n, t, 1, gz),
             fixed_fastq(os.path.join(tmp, "I1.fastq"), n, bcs[:, :8], 1, gz),
             fixed_fastq(os.path.join(tmp, "I2.fastq"), n, bcs[:, 8:16], 2, gz),
             fixed_fastq(os.path.join(tmp, "R2.fastq"), n, t[::-1].copy(), 2, gz)]
    in_bytes = sum(os.path.getsize(f) for f in files)
    meta = os.path.join(tmp, "meta.tsv")
    with open(meta, "w") as fh:
        fh.write("sample_id\tbarcode\n" + "".join(f"S{i:04}\t{b}\n" for i, b in enumerate(w.barcodes)))
    out = os.path.join(tmp, "out")
    exe = os.path.join(ROOT, "fqtk_amd", "bin", "fqtk")
    cmd = [exe, "demux", "-i", *files, "-r", "150T", "8B", "8B", "150T", "-s", meta, "-o", out, "-t", str(threads),
           ]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FQTK_TIMING="1"))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split("\t") for l in open(os.path.join(out, "demux-metrics.txt")).read().splitlines()[1:]]
    assert sum(int(x[2]) for x in rows) == n
    out_bytes = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    return {"templates": n, "threads": threads, "gz_inputs": gz, "seconds": round(dt, 3),
            "M_templates_per_s": round(n / dt / 1e6, 3), "input_MB": round(in_bytes / 1e6, 1),
            "output_MB": round(out_bytes / 1e6, 1), "host_cores": os.cpu_count(),
            "log": [l for l in r.stderr.splitlines() if "INFO" in l and "demultiplexed" not in l]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--templates", type=int, default=2_000_000)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--skip-b", action="store_true")
    a = ap.parse_args()
    res = {}
    if not a.skip_b:
        res["scope_B_boundary"] = scope_b()
    tmp = tempfile.mkdtemp(prefix="fqtk_e2e_", dir="/tmp")
    try:
        res["scope_E_cli"] = scope_e(a.templates, a.threads, a.gz, tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(res))
