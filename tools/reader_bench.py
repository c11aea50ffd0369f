#!/usr/bin/env python3
"""Developer tool: the input side alone -- decode + parse of one FASTQ file through the reader of `fqtk demux` --
for a plain file, its gzip and BGZF forms, with 1 .. n decoder threads per gzip stream.
    python tools/reader_bench.py [--records 8000000]"""
import argparse
import ctypes as C
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import hostlib as H  # noqa: E402
from tests.test_host_components import _fastq_like  # noqa: E402


def count(path, gz_threads=0, batch=131072):
    fn = H.lib().fqtk_host_fastq_count
    fn.restype = C.c_int64
    err = C.create_string_buffer(256)
    b = C.c_uint64(0)
    t = time.time()
    n = fn(path.encode(), C.c_uint64(batch), C.c_uint32(2), C.c_uint32(gz_threads), C.byref(b), err, C.c_size_t(256))
    dt = time.time() - t
    assert n >= 0, err.value
    return n, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=8_000_000)
    a = ap.parse_args()
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="fqtk_rb_", dir=base)
    block = _fastq_like(100000, 1)
    reps = max(1, a.records // 100000)
    plain = os.path.join(tmp, "r.fq")
    with open(plain, "wb") as f:
        for _ in range(reps):
            f.write(block)
    size = os.path.getsize(plain)
    gz = plain + ".gz"
    subprocess.check_call(f"gzip -1 -c {plain} > {gz}", shell=True)
    try:
        n, dt = count(plain)
        print(f"plain (mapped, parsed in place): {n / dt / 1e6:.2f} M records/s, {size / dt / 1e9:.2f} GB/s")
        for t in (1, 2, 4, 5, 6, 8):
            n, dt = count(gz, gz_threads=t)
            print(f"gzip, {t} decoder thread(s): {n / dt / 1e6:.2f} M records/s, {size / dt / 1e9:.2f} GB/s of text")
    finally:
        for p in (plain, gz):
            os.remove(p)
        os.rmdir(tmp)


if __name__ == "__main__":
    main()
