#!/usr/bin/env python3
"""Soak test of `fqtk demux` (run on the GPU box): random read structures / inputs / chunking / threads,
checked against the CPU oracle's assignments and an independent Python segment extraction:
  * every non-skipped template appears exactly once, in the file set of the sample the oracle assigns;
  * within each output file records keep input order; bases/quals are the right segment;
  * the header carries the read number, the '+'-joined sample barcodes and (if any) the UMIs;
  * demux-metrics.txt counts equal the oracle's.
usage: python tools/soak_cli.py [--iters 40] [--seed 1]"""
import argparse
import gzip
import os
import random
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

EXE = os.path.join(ROOT, "fqtk_amd", "bin", "fqtk")
CODES = {"T": "R", "B": "I", "M": "U", "C": "C"}


def parse_rs(rs):
    segs, off = [], 0
    for m in re.finditer(r"(\d+|\+)([TBMCS])", rs):
        ln = None if m.group(1) == "+" else int(m.group(1))
        segs.append((off, ln, m.group(2)))
        off += ln or 0
    return segs


def read_fq(path):
    with gzip.open(path, "rt") as fh:
        lines = fh.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    return [(lines[i][1:], lines[i + 1], lines[i + 3]) for i in range(0, len(lines), 4)]


EXTRA = []   # extra fqtk demux arguments of this run


def one(rng, it, tmp):
    n_inputs = rng.randint(1, 4)
    kinds_pool = ["T", "B", "M", "C", "S"]
    structures = []
    for _ in range(n_inputs):
        nseg = rng.randint(1, 4)
        segs = [(rng.randint(1, 9), rng.choice(kinds_pool)) for _ in range(nseg)]
        variable_last = rng.random() < 0.3
        rs = "".join(f"{l}{k}" for l, k in segs[:-1]) + (f"+{segs[-1][1]}" if variable_last else f"{segs[-1][0]}{segs[-1][1]}")
        structures.append(rs)
    if not any("B" in s for s in structures):
        structures[0] = "6B" + structures[0]
    if not any("T" in s for s in structures):
        structures[-1] = structures[-1] + "" if structures[-1].endswith("T") else "5T" if False else structures[-1]
    parsed = [parse_rs(s) for s in structures]
    # fixed total barcode length required for a sensible table; variable B segments get length 1..6 per read
    n = rng.randint(1, 20000)
    nprng = np.random.default_rng(rng.randint(0, 1 << 30))
    S = rng.randint(1, 40)
    fixed_L = sum(l for p in parsed for (_, l, k) in p if k == "B" and l is not None)
    var_b = [(i, j) for i, p in enumerate(parsed) for j, (_, l, k) in enumerate(p) if k == "B" and l is None]
    L = fixed_L + (3 if var_b else 0)      # variable B segments mostly carry 3 bases
    if L == 0 or L > 40:
        return "skipped-degenerate"
    barcodes = set()
    sample_alphabet = list("ACGT") if rng.random() < 0.7 else list("ACGTACGTACGTNRYKMSWBDHV")   # 30 %: IUPAC-degenerate tables
    read_noise = "ACGTN" if rng.random() < 0.5 else "ACGTNACGTNn.RYKMuX#"                        # 50 %: odd bytes in the reads
    while len(barcodes) < S:
        barcodes.add("".join(nprng.choice(sample_alphabet, size=L)))
        if len(barcodes) < S and 4 ** L <= len(barcodes):
            break
    barcodes = sorted(barcodes)
    S = len(barcodes)
    reads = [[None] * n for _ in range(n_inputs)]
    skip_ok = rng.random() < 0.5
    short_any = False
    for t in range(n):
        src = barcodes[nprng.integers(0, S)] if nprng.random() < 0.85 else "".join(nprng.choice(list("ACGTN"), size=L))
        src = "".join(c if nprng.random() > 0.03 else read_noise[nprng.integers(0, len(read_noise))] for c in src)
        pos = 0
        for i, p in enumerate(parsed):
            s = ""
            for (_, ln, k) in p:
                if k == "B":
                    take = ln if ln is not None else (3 if nprng.random() < 0.9 else int(nprng.integers(1, 6)))
                    piece = src[pos:pos + take]
                    piece = piece + "A" * (take - len(piece))
                    pos += ln if ln is not None else 3
                    s += piece
                else:
                    take = ln if ln is not None else int(nprng.integers(1, 12))
                    s += "".join(nprng.choice(list("ACGT"), size=take))
            if skip_ok and nprng.random() < 0.01:
                s = s[: max(0, len(s) - int(nprng.integers(1, 4)))]
            reads[i][t] = s
    files = []
    # bgzf: members inflated on the device (fqtk_demuxer_feed), cut into chunks by line counts; mixed: a BGZF file with ordinary gzip members
    # in it (`cat a.bgz b.gz`): the feeder decodes those as serial streams, in chunks, between its runs of BGZF members
    kind = rng.choice(["plain", "gz", "gz", "bgzf", "bgzf", "mixed"])
    gz = kind == "gz"
    member = rng.choice([300, 4000, 65280])            # text bytes per BGZF member: members and chunks never line up
    for i in range(n_inputs):
        path = os.path.join(tmp, f"in{it}_{i}.fastq" + (".gz" if kind != "plain" else ""))
        gz = kind == "gz"
        text = "".join(f"@q_{t} {i + 1}:N:0:0\n{reads[i][t]}\n+\n{'I' * len(reads[i][t])}\n" for t in range(n))
        if kind in ("bgzf", "mixed"):
            import struct
            import zlib
            raw = text.encode()
            if raw and rng.random() < 0.3:
                raw = raw[:-1]                          # last line without a newline
            data = b""
            o = 0
            while o < len(raw):
                if kind == "mixed" and o and rng.random() < 0.15:   # an ordinary gzip member of up to 40 KB of text (any level; a sync flush now and then)
                    piece = raw[o:o + rng.randint(1, 40000)]
                    c = zlib.compressobj(rng.choice([0, 1, 6, 9]), zlib.DEFLATED, 31)
                    half = len(piece) // 2
                    data += c.compress(piece[:half]) + (c.flush(zlib.Z_SYNC_FLUSH) if rng.random() < 0.5 else b"") + c.compress(piece[half:]) + c.flush()
                    o += len(piece)
                    continue
                piece = raw[o:o + member]
                c = zlib.compressobj(rng.choice([0, 1, 6, 9]), zlib.DEFLATED, -15)
                payload = c.compress(piece) + c.flush()
                data += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(payload) + 8 - 1) + payload +
                         struct.pack("<II", zlib.crc32(piece), len(piece)))
                o += len(piece)
            data += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
            open(path, "wb").write(data)
        else:
            (gzip.open(path, "wt") if gz else open(path, "w")).write(text)
        files.append(path)
    meta = os.path.join(tmp, f"meta{it}.tsv")
    open(meta, "w").write("sample_id\tbarcode\n" + "".join(f"S{i}\t{b}\n" for i, b in enumerate(barcodes)))
    out = os.path.join(tmp, f"out{it}")
    types = [k for k in "TBMC" if rng.random() < 0.6 and any(k in s for s in structures)] or ["B"]
    mm, delta = rng.choice([0, 1, 1, 2]), rng.choice([0, 1, 2, 2])
    cmd = [EXE, "demux", "-i", *files, "-r", *structures, "-s", meta, "-o", out, "-b", *types, "--max-mismatches", str(mm),
           "-d", str(delta), "-t", str(rng.randint(5, 16)), "--chunk-reads", str(rng.choice([1, 7, 100, 1000, 5000, 262144]))]
    if skip_ok:
        cmd += ["-S", "too-few-bases"]
    cmd += EXTRA
    if EXE.endswith(".thread") and shutil.which("setarch"):   # TSan's shadow layout does not survive this kernel's ASLR range
        cmd = ["setarch", os.uname().machine, "-R"] + cmd
    env = dict(os.environ)
    if kind == "gz" and rng.random() < 0.7:
        env["FQTK_GPU_GUNZIP"] = "1"                                  # serial gzip decoded on the device in chunks
        env["FQTK_GZ_DEVICE_CHUNK_KB"] = str(rng.choice([4, 16, 64, 512]))
        env["FQTK_GZ_DEVICE_CHUNKS"] = str(rng.choice([3, 20, 448]))
        if rng.random() < 0.5:
            env["FQTK_FED_ARENA_MIN"] = str(rng.choice([20000, 300000]))
    if kind in ("gz", "mixed"):
        if rng.random() < 0.3:
            env["FQTK_GZ_FORCE_FALLBACK"] = str(rng.choice([1, 2, 3]))   # every k-th stretch by the host's sequential decoder
        if rng.random() < 0.3:
            env["FQTK_GZ_DEVICE_SYMS"] = str(rng.choice([1, 2]))         # chunks run out of room for symbols and end at a block boundary
        if kind == "mixed":
            env["FQTK_GZ_DEVICE_CHUNK_KB"] = str(rng.choice([4, 16, 64]))
            env["FQTK_GZ_DEVICE_CHUNKS"] = str(rng.choice([2, 20, 448]))
    if kind in ("bgzf", "mixed") and rng.random() < 0.5:
        env["FQTK_FED_ARENA_MIN"] = str(rng.choice([20000, 300000]))   # the fed text changes arena every few chunks
    if EXE.endswith(".thread"):
        env["TSAN_OPTIONS"] = "suppressions=" + os.path.join(ROOT, "tools", "tsan.supp") + ":report_signal_unsafe=0:second_deadlock_stack=1"
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    for mark in ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "runtime error:"):   # (a sanitizer build: --exe)
        if mark in r.stderr:
            at = r.stderr.index(mark)
            raise AssertionError("sanitizer report from " + " ".join(cmd) + "\n" + r.stderr[at:at + 4000])
    # expected
    minlen = [sum((l if l is not None else 1) for (_, l, _) in p) for p in parsed]
    skipped = [any(len(reads[i][t]) < minlen[i] for i in range(n_inputs)) for t in range(n)]
    if any(skipped) and not skip_ok:
        assert r.returncode != 0 and "had too few bases" in r.stderr, (cmd, r.stderr[-500:])
        return "fatal-short-ok"

    def seg(i, j, t):
        off, ln, _ = parsed[i][j]
        s = reads[i][t]
        return s[off:off + ln] if ln is not None else s[off:]
    bsegs = [[seg(i, j, t) for i, p in enumerate(parsed) for j, (_, _, k) in enumerate(p) if k == "B"] for t in range(n)]
    obs = ["".join(b) for b in bsegs]
    lit = O.RefLiteral(barcodes, mm, delta, True)
    assign, too_long = [], False
    for t in range(n):
        if skipped[t]:
            assign.append(None)
            continue
        try:
            a = lit.assign(obs[t].encode())
        except O.OracleLengthError:
            too_long = True
            break
        assign.append(S if a is None else a[0])
    if too_long:
        # (the reference formats the read barcode into that sentence with decode(), which panics first --
        #  "Invalid bit mask for base: 0", mod.rs:80 -- when the read holds a byte without an IUPAC mask)
        assert r.returncode != 0 and ("differs from expected barcode (" in r.stderr or "Invalid bit mask for base: 0" in r.stderr), (cmd, r.stderr[-500:])
        return "fatal-long-ok"
    assert r.returncode == 0, (cmd, r.stderr[-800:])
    names = [f"S{i}" for i in range(S)] + ["unmatched"]
    counts = [sum(1 for a in assign if a == s) for s in range(S + 1)]
    rows = [l.split("\t") for l in open(os.path.join(out, "demux-metrics.txt")).read().splitlines()[1:]]
    assert [int(x[2]) for x in rows] == counts, (cmd, counts[:5])
    umis = [[seg(i, j, t) for i, p in enumerate(parsed) for j, (_, _, k) in enumerate(p) if k == "M"] for t in range(n)]
    for k in types:
        refs = [(i, j) for i, p in enumerate(parsed) for j, (_, _, kk) in enumerate(p) if kk == k]
        for num, (i, j) in enumerate(refs, start=1):
            for s, name in enumerate(names):
                recs = read_fq(os.path.join(out, f"{name}.{CODES[k]}{num}.fq.gz"))
                exp_t = [t for t in range(n) if assign[t] == s]
                assert len(recs) == len(exp_t), (cmd, name, k, num)
                for (h, sq, ql), t in zip(recs, exp_t):
                    nm = f"q_{t}" + (":" + "+".join(umis[t]) if umis[t] else "")
                    assert h == f"{nm} {num}:N:0:" + "+".join(bsegs[t]), (cmd, h, t)
                    assert sq == seg(i, j, t) and ql == "I" * len(sq), (cmd, name, t)
    return "ok"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--exe", default="", help="another build of the binary, e.g. fqtk_amd/bin/fqtk.thread (python -m fqtk_amd.build --sanitize=thread): "
                                              "its stderr is scanned for sanitizer reports")
    ap.add_argument("--host-output", action="store_true", help="every run with --host-output (records formatted and compressed by the host threads; default: on the device)")
    ap.add_argument("--devices", default="", help="every run with --devices <this> (e.g. 0,0: two record pipelines on one GPU; compressed inputs at their home pipelines)")
    a = ap.parse_args()
    if a.exe:
        EXE = os.path.abspath(a.exe)
    EXTRA[:] = (["--host-output"] if a.host_output else []) + (["--devices", a.devices] if a.devices else [])
    rng = random.Random(a.seed)
    tmp = tempfile.mkdtemp(prefix="fqtk_soak_", dir="/tmp")
    tally = {}
    try:
        for it in range(a.iters):
            res = one(rng, it, tmp)
            tally[res] = tally.get(res, 0) + 1
            shutil.rmtree(os.path.join(tmp, f"out{it}"), ignore_errors=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("soak_cli", tally)
