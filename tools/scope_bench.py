#!/usr/bin/env python3
"""Scopes B and E of SURVEY.md 8(d) (scope K, HBM-resident, is bench.py's `value`):
  B  boundary level: pinned host SoA barcodes -> fqtk_matcher_enqueue/wait -> host results (PCIe incl.)
  E  end to end: the `fqtk demux` binary on synthetic dual-index FASTQ files (plain or gz) -> per-sample
     BGZF files + metrics; covers what Demux::execute covers (/root/reference/src/bin/commands/demux.rs:881-1001)
bench.py imports scope_b()/scope_e() and prints them inside its JSON line (`scopes`); standalone:
    python tools/scope_bench.py [--templates 4000000] [--threads 32] [--gz] [--skip-b]
prints one JSON object.  Neither number is ever bench.py's `value`."""
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


def scope_b(cfg_id=3, n_chunk=8_000_000, chunks=12, matcher=None, workload=None):
    """Two pinned slots, `chunks` chunks of n_chunk reads: H2D of chunk k+1 overlaps kernel + D2H of chunk k."""
    cfg = synth.CONFIGS[cfg_id]
    w = workload or synth.Workload(cfg)
    lib = _lib.load()
    m = matcher or BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    bufs = []
    for s in range(2):
        po, pr = C.c_void_p(), C.c_void_p()
        assert lib.fqtk_pinned_alloc(n_chunk * cfg.stride, C.byref(po)) == 0, _lib.last_error()
        assert lib.fqtk_pinned_alloc(n_chunk * 4, C.byref(pr)) == 0, _lib.last_error()
        host = w.fill_host(s * n_chunk, n_chunk)
        C.memmove(po, host.ctypes.data, host.nbytes)
        bufs.append((po, pr))
    try:
        for s in range(2):   # warm-up (staging buffers are allocated on first use)
            assert lib.fqtk_matcher_enqueue(m.handle, s, bufs[s][0], cfg.stride, None, n_chunk, bufs[s][1]) == 0
        for s in range(2):
            assert lib.fqtk_matcher_wait(m.handle, s) == 0
        dt = float("inf")
        for _rep in range(3):   # (a pass is 20-30 ms: the best of three, a hiccup of the host costs a third of one)
            t0 = time.perf_counter()
            for c in range(chunks):
                s = c % 2
                if c >= 2:
                    assert lib.fqtk_matcher_wait(m.handle, s) == 0
                assert lib.fqtk_matcher_enqueue(m.handle, s, bufs[s][0], cfg.stride, None, n_chunk, bufs[s][1]) == 0
            for s in range(2):
                assert lib.fqtk_matcher_wait(m.handle, s) == 0
            dt = min(dt, time.perf_counter() - t0)
        # the results that came back are the matcher's: spot-check one slot against the sync path
        got = np.ctypeslib.as_array(C.cast(bufs[1][1], C.POINTER(C.c_uint32)), (n_chunk,))[:100_000].copy()
        ref, _ = m.assign_batch(w.fill_host(n_chunk, 100_000), counts=False)
        assert np.array_equal(got, ref.view(np.uint32)), "scope B results differ from the synchronous path"
        scratch = np.zeros(cfg.n_samples + 1, dtype=np.uint64)
        lib.fqtk_matcher_counts(m.handle, scratch.ctypes.data)   # leave the handle's accumulator empty
    finally:
        for po, pr in bufs:
            lib.fqtk_pinned_free(po)
            lib.fqtk_pinned_free(pr)
    reads = n_chunk * chunks
    return {"what": "fqtk_matcher_enqueue/wait on 2 pinned slots: host SoA barcodes -> host results, PCIe inclusive",
            "workload": cfg.name, "reads": reads, "chunk_reads": n_chunk, "seconds": round(dt, 4),
            "M_reads_per_s": round(reads / dt / 1e6, 1),
            "GB_per_s_over_pcie": round(reads * (cfg.stride + 4) / dt / 1e9, 2)}


def scope_b_devices(cfg_id, devices, n_chunk=8_000_000, chunks=12, workload=None):
    """Scope B over several devices ("report per-GPU-count reads/sec for all three scopes"): one matcher per entry of
    `devices` (table replicated), each driven by its own host thread over its own two page-locked slots -- the reads are sharded,
    nothing is exchanged.  All threads start together; reads of all devices / the time until the last one is done."""
    import threading
    cfg = synth.CONFIGS[cfg_id]
    w = workload or synth.Workload(cfg)
    lib = _lib.load()
    G = len(devices)
    ms = [BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, device=d) for d in devices]
    host = [w.fill_host(s * n_chunk, n_chunk) for s in range(2)]
    bufs = []
    for g in range(G):
        per = []
        for s in range(2):
            po, pr = C.c_void_p(), C.c_void_p()
            assert lib.fqtk_pinned_alloc(n_chunk * cfg.stride, C.byref(po)) == 0, _lib.last_error()
            assert lib.fqtk_pinned_alloc(n_chunk * 4, C.byref(pr)) == 0, _lib.last_error()
            C.memmove(po, host[s].ctypes.data, host[s].nbytes)
            per.append((po, pr))
        bufs.append(per)
    errors = []

    def one_pass(g, n_chunks, barrier):
        try:
            m = ms[g]
            if barrier is not None:
                barrier.wait()
            for c in range(n_chunks):
                s = c % 2
                if c >= 2:
                    assert lib.fqtk_matcher_wait(m.handle, s) == 0, _lib.last_error()
                assert lib.fqtk_matcher_enqueue(m.handle, s, bufs[g][s][0], cfg.stride, None, n_chunk, bufs[g][s][1]) == 0, _lib.last_error()
            for s in range(min(2, n_chunks)):
                assert lib.fqtk_matcher_wait(m.handle, s) == 0, _lib.last_error()
        except BaseException as e:   # noqa: BLE001 -- reported by the caller
            errors.append(e)
            if barrier is not None:
                barrier.abort()

    try:
        for g in range(G):
            one_pass(g, 2, None)   # warm-up (staging buffers are made on first use)
        assert not errors, errors
        dt = float("inf")
        for _rep in range(3):
            barrier = threading.Barrier(G + 1)
            ts = [threading.Thread(target=one_pass, args=(g, chunks, barrier)) for g in range(G)]
            for t in ts:
                t.start()
            barrier.wait()
            t0 = time.perf_counter()
            for t in ts:
                t.join()
            dt = min(dt, time.perf_counter() - t0)
            assert not errors, errors
        ref, _ = ms[0].assign_batch(w.fill_host(n_chunk, 100_000), counts=False)
        for g in range(G):   # every device's results are the matcher's
            got = np.ctypeslib.as_array(C.cast(bufs[g][1][1], C.POINTER(C.c_uint32)), (n_chunk,))[:100_000].copy()
            assert np.array_equal(got, ref.view(np.uint32)), f"scope B results of device {devices[g]} differ from the synchronous path"
    finally:
        for per in bufs:
            for po, pr in per:
                lib.fqtk_pinned_free(po)
                lib.fqtk_pinned_free(pr)
        for m in ms:
            m.close()
    reads = n_chunk * chunks * G
    return {"what": f"{G} matchers (devices {','.join(map(str, devices))}), a host thread and 2 pinned slots each: host SoA barcodes -> host results, PCIe inclusive",
            "workload": cfg.name, "devices": list(devices), "reads": reads, "chunk_reads": n_chunk, "seconds": round(dt, 4),
            "M_reads_per_s": round(reads / dt / 1e6, 1), "GB_per_s_over_pcie": round(reads * (cfg.stride + 4) / dt / 1e9, 2)}


def scope_b_packed(cfg_id=3, n_chunk=8_000_000, chunks=12, matcher=None, workload=None, threads=0):
    """Scope B with 4 bits per base over the link (fqtk_pack_barcodes + fqtk_matcher_enqueue_packed).  The PACKER RUNS INSIDE
    THE CLOCK (a rate the box cannot feed is not a rate): every chunk's ASCII rows are packed by `threads` host
    threads into the slot's page-locked buffer, then enqueued; packing chunk c + 1 overlaps the link carrying chunk c.  The
    packer's own rate on one thread is reported beside (SSSE3: one pshufb validates 16 bases, one pmaddubsw packs them)."""
    from concurrent.futures import ThreadPoolExecutor
    cfg = synth.CONFIGS[cfg_id]
    w = workload or synth.Workload(cfg)
    lib = _lib.load()
    m = matcher or BarcodeMatcher(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta)
    ps = int(lib.fqtk_packed_stride(cfg.barcode_len))
    T = threads or max(1, min(16, len(os.sched_getaffinity(0))))
    ascii_rows = [w.fill_host(s * n_chunk, n_chunk) for s in range(2)]      # the caller's SoA rows (as scope B's)
    bufs = []
    for s in range(2):
        pp, pr = C.c_void_p(), C.c_void_p()
        assert lib.fqtk_pinned_alloc(n_chunk * ps, C.byref(pp)) == 0, _lib.last_error()
        assert lib.fqtk_pinned_alloc(n_chunk * 4, C.byref(pr)) == 0, _lib.last_error()
        bufs.append((pp, pr))
    cap = 1 << 16
    cuts = [(n_chunk * t // T, n_chunk * (t + 1) // T) for t in range(T)]
    exc = [[(np.empty(cap, dtype=np.uint32), np.empty((cap, cfg.barcode_len), dtype=np.uint8)) for _ in range(T)] for _ in range(2)]

    def pack_slice(s, t):
        lo, hi = cuts[t]
        k = C.c_uint64(0)
        ei, er = exc[s][t]
        rc = lib.fqtk_pack_barcodes(ascii_rows[s].ctypes.data + lo * cfg.stride, cfg.stride, cfg.barcode_len, hi - lo,
                                    bufs[s][0].value + lo * ps, ps, ei.ctypes.data, er.ctypes.data, cap, C.byref(k))
        assert rc == 0, _lib.last_error()
        return int(k.value)

    pool = ThreadPoolExecutor(T)

    def pack_and_enqueue(s):
        ks = list(pool.map(lambda t: pack_slice(s, t), range(T)))
        n_exc = sum(ks)
        if n_exc:   # the slices' exceptions, their indices rebased to the chunk
            ei = np.concatenate([exc[s][t][0][:ks[t]] + np.uint32(cuts[t][0]) for t in range(T)])
            er = np.concatenate([exc[s][t][1][:ks[t]] for t in range(T)])
        else:
            ei = er = None
        rc = lib.fqtk_matcher_enqueue_packed(m.handle, s, bufs[s][0], ps, n_chunk, ei.ctypes.data if n_exc else None,
                                             er.ctypes.data if n_exc else None, n_exc, bufs[s][1])
        assert rc == 0, _lib.last_error()
        return ei, er   # (kept alive until the wait)

    try:
        t0 = time.perf_counter()
        pack_slice(0, 0)
        one_thread = (cuts[0][1] - cuts[0][0]) / (time.perf_counter() - t0)
        keep = [pack_and_enqueue(s) for s in range(2)]
        for s in range(2):
            assert lib.fqtk_matcher_wait(m.handle, s) == 0
        dt = float("inf")
        for _rep in range(3):
            t0 = time.perf_counter()
            for c in range(chunks):
                s = c % 2
                if c >= 2:
                    assert lib.fqtk_matcher_wait(m.handle, s) == 0
                keep[s] = pack_and_enqueue(s)
            for s in range(2):
                assert lib.fqtk_matcher_wait(m.handle, s) == 0
            dt = min(dt, time.perf_counter() - t0)
        got = np.ctypeslib.as_array(C.cast(bufs[1][1], C.POINTER(C.c_uint32)), (n_chunk,))[:100_000].copy()
        ref, _ = m.assign_batch(w.fill_host(n_chunk, 100_000), counts=False)
        assert np.array_equal(got, ref.view(np.uint32)), "packed scope B results differ from the synchronous ASCII path"
        scratch = np.zeros(cfg.n_samples + 1, dtype=np.uint64)
        lib.fqtk_matcher_counts(m.handle, scratch.ctypes.data)
    finally:
        pool.shutdown()
        for pp, pr in bufs:
            lib.fqtk_pinned_free(pp)
            lib.fqtk_pinned_free(pr)
    reads = n_chunk * chunks
    return {"what": "host ASCII rows -> fqtk_pack_barcodes on `pack_threads` host threads (INSIDE the clock) -> fqtk_matcher_enqueue_packed/wait "
                    "on 2 pinned slots -> host results, PCIe inclusive",
            "workload": cfg.name, "reads": reads, "chunk_reads": n_chunk, "packed_bytes_per_read": ps, "seconds": round(dt, 4), "passes": 3,
            "pack_threads": T, "M_reads_per_s": round(reads / dt / 1e6, 1), "GB_per_s_over_pcie": round(reads * (ps + 4) / dt / 1e9, 2),
            "host_packer_M_reads_per_s_1_thread": round(one_thread / 1e6, 1)}


def fixed_fastq(path, start, n, seqs, read_no, append):
    """n records with fixed-width fields, built as one numpy byte matrix; names carry start + row."""
    L = seqs.shape[1]
    head = np.frombuffer(b"@inst:1:FC:1:0000000000 %d:N:0:0\n" % read_no, dtype=np.uint8)
    rec = np.empty((n, len(head) + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, :len(head)] = head
    digits = (start + np.arange(n))[:, None] // (10 ** np.arange(9, -1, -1))[None, :] % 10
    rec[:, 13:23] = digits.astype(np.uint8) + ord("0")
    o = len(head)
    rec[:, o:o + L] = seqs
    rec[:, o + L] = ord("\n")
    rec[:, o + L + 1] = ord("+")
    rec[:, o + L + 2] = ord("\n")
    rec[:, o + L + 3:o + 2 * L + 3] = ord("I")
    rec[:, -1] = ord("\n")
    with open(path, "ab" if append else "wb") as fh:
        fh.write(rec.tobytes())


def bgzf_file(path, level=1):
    """path -> path.gz as BGZF (blocks of 65280 bytes + EOF marker) through the host shim's block compressor."""
    lib = C.CDLL(os.path.join(ROOT, "fqtk_amd", "lib", "libfqtk_host.so"))
    with open(path, "rb") as fi, open(path + ".gz", "wb") as fo:
        while True:
            data = fi.read(64 * 65280)
            if not data:
                break
            cap = len(data) + len(data) // 8 + 65536
            out = (C.c_uint8 * cap)()
            n = C.c_size_t()
            assert lib.fqtk_host_bgzf(data, C.c_size_t(len(data)), level, out, C.c_size_t(cap), C.byref(n)) == 0
            fo.write(bytes(out[:n.value - 28]))          # drop the per-call EOF marker
        fo.write(bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    os.unlink(path)
    return path + ".gz"


def make_inputs(tmp, n, gz, block=1_000_000, repeat_first_block=False):
    """cfg 3's shape as files: R1 150T, I1 8B, I2 8B, R2 150T; barcodes from the same synthetic stream as
    scopes K/B; template bases are one random 1 M x 150 block reused per block (names differ).
    repeat_first_block: every block is a byte copy of the first (same names, same barcodes) -- large inputs
    in seconds instead of minutes; the per-sample counts are then (n / block) x the first block's."""
    cfg = synth.CONFIGS[3]
    w = synth.Workload(cfg)
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    t1 = acgt[rng.integers(0, 4, size=(min(block, n), 150))]
    t2 = t1[::-1].copy()
    names = ["R1.fastq", "I1.fastq", "I2.fastq", "R2.fastq"]
    paths = [os.path.join(tmp, x) for x in names]
    if repeat_first_block:
        assert n % block == 0
        bcs = w.fill_host(0, block)
        for path, seqs, rn in ((paths[0], t1, 1), (paths[1], bcs[:, :8], 1), (paths[2], bcs[:, 8:16], 2), (paths[3], t2, 2)):
            fixed_fastq(path, 0, block, seqs, rn, False)
            blob = open(path, "rb").read()
            with open(path, "ab") as fh:
                for _ in range(n // block - 1):
                    fh.write(blob)
    for lo in range(0, 0 if repeat_first_block else n, block):
        cur = min(block, n - lo)
        bcs = w.fill_host(lo, cur)
        fixed_fastq(paths[0], lo, cur, t1[:cur], 1, lo > 0)
        fixed_fastq(paths[1], lo, cur, bcs[:, :8], 1, lo > 0)
        fixed_fastq(paths[2], lo, cur, bcs[:, 8:16], 2, lo > 0)
        fixed_fastq(paths[3], lo, cur, t2[:cur], 2, lo > 0)
    if gz == "bgzf":
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(4) as ex:
            paths = list(ex.map(bgzf_file, paths))
    elif gz:
        procs = [subprocess.Popen(["gzip", "-1", "-f", p]) for p in paths]
        assert all(p.wait() == 0 for p in procs)
        paths = [p + ".gz" for p in paths]
    meta = os.path.join(tmp, "meta.tsv")
    with open(meta, "w") as fh:
        fh.write("sample_id\tbarcode\n" + "".join(f"S{i:04}\t{b}\n" for i, b in enumerate(w.barcodes)))
    return paths, meta, w


def prefix_inputs(tmp, paths, frac_num, frac_den, sub="sub"):
    """The first frac_num / frac_den of every (fixed-width-record, plain) input, copied next to it."""
    d = os.path.join(tmp, sub)
    os.makedirs(d, exist_ok=True)
    out = []
    for p in paths:
        size = os.path.getsize(p) * frac_num // frac_den
        q = os.path.join(d, os.path.basename(p))
        with open(p, "rb") as fi, open(q, "wb") as fo:
            left = size
            while left:
                b = fi.read(min(left, 64 << 20))
                fo.write(b)
                left -= len(b)
        out.append(q)
    return out


def _gzip_repeated(args):
    path, block_bytes, reps = args
    import zlib as _z
    size = os.path.getsize(path)
    assert size % block_bytes == 0
    with open(path, "rb") as fh:
        block = fh.read(block_bytes)
    c = _z.compressobj(1, _z.DEFLATED, -15)
    body = c.compress(block) + c.flush(_z.Z_FULL_FLUSH)      # whole deflate blocks, byte-aligned, window reset: repeatable
    reps = reps or size // block_bytes
    size = reps * block_bytes
    crc = 0
    for _ in range(reps):
        crc = _z.crc32(block, crc)
    with open(path + ".gz", "wb") as fo:
        fo.write(bytes([0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3]))
        for _ in range(reps):
            fo.write(body)
        fo.write(bytes([0x03, 0x00]))                                # final empty block (fixed Huffman, end of block)
        fo.write(crc.to_bytes(4, "little") + (size & 0xFFFFFFFF).to_bytes(4, "little"))
    return path + ".gz"


def _bgzf_repeated(args):
    path, block_bytes, reps = args
    size = os.path.getsize(path)
    assert size % block_bytes == 0
    reps = reps or size // block_bytes
    with open(path, "rb") as fh:
        block = fh.read(block_bytes)
    lib = C.CDLL(os.path.join(ROOT, "fqtk_amd", "lib", "libfqtk_host.so"))
    body = bytearray()
    for o in range(0, len(block), 64 * 65280):
        data = block[o:o + 64 * 65280]
        cap = len(data) + len(data) // 8 + 65536
        out = (C.c_uint8 * cap)()
        n = C.c_size_t()
        assert lib.fqtk_host_bgzf(data, C.c_size_t(len(data)), 1, out, C.c_size_t(cap), C.byref(n)) == 0
        body += bytes(out[:n.value - 28])                            # without the per-call EOF marker
    with open(path + ".bgz", "wb") as fo:
        for _ in range(reps):
            fo.write(body)
        fo.write(bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    return path + ".bgz"


def bgzf_repeated(paths, block_records=1_000_000, reps=None):
    """path -> path.bgz: BGZF (members of 65 280 bytes of text, level 1, EOF marker: what bgzip writes) of a file that
    repeats its first block_records records: the block's members are compressed once and written size / block times
    (reps given: the file IS one block, and the BGZF file holds it reps times -- inputs larger than the scratch could hold as text)."""
    from concurrent.futures import ProcessPoolExecutor
    jobs = []
    for p in paths:
        with open(p, "rb") as fh:
            head = fh.read(1 << 16)
        rec = head.index(b"\n", head.index(b"\n", head.index(b"\n", head.index(b"\n") + 1) + 1) + 1) + 1   # fixed-width records
        jobs.append((p, rec * block_records, reps))
    with ProcessPoolExecutor(4) as ex:
        return list(ex.map(_bgzf_repeated, jobs))


def gzip_single_stream(paths, block_records=1_000_000, reps=None):
    """path -> path.gz: ONE gzip member per file (one serial DEFLATE stream, level 1, what `gzip -1` / bcl2fastq write)
    of a file that repeats its first block_records records: the block is compressed once and its deflate blocks are
    repeated inside the stream, which takes seconds instead of the minute `gzip -1` needs for 12 GB
    (reps given: the file IS one block, and the stream holds it reps times)."""
    from concurrent.futures import ProcessPoolExecutor
    jobs = []
    for p in paths:
        with open(p, "rb") as fh:
            head = fh.read(1 << 16)
        rec = head.index(b"\n", head.index(b"\n", head.index(b"\n", head.index(b"\n") + 1) + 1) + 1) + 1   # fixed-width records
        jobs.append((p, rec * block_records, reps))
    with ProcessPoolExecutor(4) as ex:
        return list(ex.map(_gzip_repeated, jobs))


def scope_e(n, threads, gz, tmp, expect_counts=None, extra_args=(), repeat_first_block=False, reuse_inputs=False, inputs=None, out_name="out"):
    if inputs is not None:   # (paths, meta) made by the caller
        paths, meta = inputs
    elif reuse_inputs:   # a second run over the files of the previous scope_e() call in the same directory
        ext = ".gz" if gz else ""
        paths = [os.path.join(tmp, x + ext) for x in ("R1.fastq", "I1.fastq", "I2.fastq", "R2.fastq")]
        meta = os.path.join(tmp, "meta.tsv")
    else:
        paths, meta, w = make_inputs(tmp, n, gz, repeat_first_block=repeat_first_block)
    in_bytes = sum(os.path.getsize(f) for f in paths)
    out = os.path.join(tmp, out_name)
    exe = os.path.join(ROOT, "fqtk_amd", "bin", "fqtk")
    cmd = [exe, "demux", "-i", *paths, "-r", "150T", "8B", "8B", "150T", "-s", meta, "-o", out, "-t", str(threads),
           *extra_args]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, FQTK_TIMING="1"))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln.split("\t") for ln in open(os.path.join(out, "demux-metrics.txt")).read().splitlines()[1:]]
    got = np.array([int(x[2]) for x in rows], dtype=np.uint64)
    assert int(got.sum()) == n
    if expect_counts is not None:     # per-sample templates of the metrics file == the oracle's count vector
        assert np.array_equal(got, expect_counts), "demux-metrics.txt differs from the oracle's per-sample counts"
    out_files = os.listdir(out)
    out_bytes = sum(os.path.getsize(os.path.join(out, f)) for f in out_files)
    peak_rss_mb = None
    for ln in r.stderr.splitlines():
        if "peak resident set" in ln:
            peak_rss_mb = float(ln.split("peak resident set ")[1].split(" MB")[0])
    is_stage = lambda ln: "thread-seconds" in ln or "main thread" in ln or "submit:" in ln or "stage seconds" in ln
    stage = [ln.split("fqtk] ", 1)[1] for ln in r.stderr.splitlines() if is_stage(ln)]
    timeline = [ln.strip() for ln in r.stderr.splitlines() if "INFO fqtk" in ln and "demultiplexed" not in ln and not is_stage(ln)]
    steady = None   # the record pipeline's own clock: first chunk submitted -> last byte written
    for ln in r.stderr.splitlines():
        if "GPU record pipeline:" in ln and "M templates/s" in ln:
            steady = float(ln.split("(")[1].split(" M templates/s")[0])
    shutil.rmtree(out, ignore_errors=True)
    return {"what": "fqtk_amd/bin/fqtk demux, files -> files, as Demux::execute demux.rs:881-1001" +
                    (": records parsed, formatted and BGZF-compressed by the host threads, barcodes matched on the GPU"
                     if "--host-output" in extra_args else
                     ": text -> GPU (records indexed, matched, formatted, DEFLATE-compressed in HBM) -> BGZF members appended"),
            "workload": "cfg3 shape: R1 150T, I1 8B, I2 8B, R2 150T; 384 samples", "templates": n, "threads": threads,
            "extra_args": list(extra_args),
            "gz_inputs": gz, "seconds": round(dt, 3), "M_templates_per_s": round(n / dt / 1e6, 3),
            "M_templates_per_s_steady": steady,
            "seconds_is": "wall clock of the command: start-up, GPU bring-up, demux, flush, every file closed, return to the caller (since round 6 the run happens in a "
                          "child process that reports when every file is closed; the kernel's clearing-up of its GPU context, 0.15-0.3 s, goes on behind the return)",
            "M_input_records_per_s": round(4 * n / dt / 1e6, 3),
            "input_MB": round(in_bytes / 1e6, 1), "output_MB": round(out_bytes / 1e6, 1), "output_files": len(out_files),
            "peak_rss_MB": peak_rss_mb,
            "files_on": tmp, "host_cores": os.cpu_count(),
            "metrics_vs_oracle": None if expect_counts is None else "per-sample counts identical", "stages": stage,
            "steady_is": "fqtk's own clock from the first chunk handed to the device to the last output byte written "
                         "(without process start, device bring-up and exit)",
            "timeline": timeline}


def scratch_dir(need_bytes):
    """RAM-backed scratch when it has room (so the number is host compute, not this box's disk), else /tmp."""
    for base in ("/dev/shm", "/tmp"):
        try:
            if shutil.disk_usage(base).free > need_bytes * 1.3:
                return tempfile.mkdtemp(prefix="fqtk_e2e_", dir=base)
        except OSError:
            pass
    return tempfile.mkdtemp(prefix="fqtk_e2e_")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--templates", type=int, default=4_000_000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--bgzf", action="store_true", help="BGZF-compress the inputs (block-parallel inflate in the reader)")
    ap.add_argument("--skip-b", action="store_true")
    ap.add_argument("--extra", default="", help="extra arguments for fqtk demux, space separated")
    ap.add_argument("--repeat-block", action="store_true", help="inputs = the first 1 M templates repeated (fast to generate)")
    a = ap.parse_args()
    res = {}
    if not a.skip_b:
        res["B"] = scope_b()
    tmp = scratch_dir(a.templates * 1100)
    try:
        res["E"] = scope_e(a.templates, a.threads, "bgzf" if a.bgzf else a.gz, tmp, extra_args=tuple(a.extra.split()),
                           repeat_first_block=a.repeat_block)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(res))
