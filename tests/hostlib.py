"""ctypes access to fqtk_amd/lib/libfqtk_host.so (C shim over the C++ host components) + CLI helpers."""
import ctypes as C
import gzip
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# FQTK_HOST_LIB: a sanitizer build of the same shim (python -m fqtk_amd.build --sanitize=address|thread; tests/test_sanitizers.py)
LIB = os.environ.get("FQTK_HOST_LIB") or os.path.join(ROOT, "fqtk_amd", "lib", "libfqtk_host.so")
EXE = os.path.join(ROOT, "fqtk_amd", "bin", "fqtk")

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.fqtk_host_parse_fastq.restype = C.c_int64
        _lib.fqtk_host_load_samples.restype = C.c_int64
        _lib.fqtk_host_format_f64.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
    return _lib


def read_structure(text):
    canon = C.create_string_buffer(256)
    err = C.create_string_buffer(512)
    min_len = C.c_uint64()
    segs = (C.c_int64 * 96)()
    n = C.c_size_t()
    rc = lib().fqtk_host_read_structure(text.encode(), canon, C.c_size_t(256), C.byref(min_len), segs,
                                        C.c_size_t(32), C.byref(n), err, C.c_size_t(512))
    if rc != 0:
        raise ValueError(err.value.decode())
    out = [(segs[3 * i], segs[3 * i + 1], chr(segs[3 * i + 2])) for i in range(n.value)]
    return canon.value.decode(), min_len.value, out


def segment_spans(text, read_len):
    n = len(read_structure(text)[2])
    spans = (C.c_uint64 * (2 * n))()
    assert lib().fqtk_host_segment_spans(text.encode(), C.c_uint64(read_len), spans, C.c_size_t(n)) == 0
    return [(spans[2 * i], spans[2 * i + 1]) for i in range(n)]


def write_header(read_num, header, bsegs, msegs):
    out = C.create_string_buffer(4096)
    err = C.create_string_buffer(512)
    b = (C.c_char_p * max(len(bsegs), 1))(*[s.encode() for s in bsegs])
    m = (C.c_char_p * max(len(msegs), 1))(*[s.encode() for s in msegs])
    rc = lib().fqtk_host_write_header(C.c_uint64(read_num), header.encode(), b, C.c_size_t(len(bsegs)), m,
                                      C.c_size_t(len(msegs)), out, C.c_size_t(4096), err, C.c_size_t(512))
    if rc != 0:
        raise ValueError(err.value.decode())
    return out.value.decode()


def parse_fastq(path, batch=1000):
    cap = 1 << 24
    out = C.create_string_buffer(cap)
    err = C.create_string_buffer(512)
    n = lib().fqtk_host_parse_fastq(str(path).encode(), C.c_uint64(batch), out, C.c_size_t(cap), err, C.c_size_t(512))
    if n < 0:
        raise ValueError(err.value.decode())
    recs = [tuple(l.split("\t")) for l in out.value.decode().split("\n") if l != ""] if n else []
    assert len(recs) == n
    return recs


def fastq_digest(path, batch=50_000, helpers=2):
    """(records, 64-bit digest of every record, kind) -- kind 0 plain, 1 gzip, 2 BGZF (block-parallel inflate)."""
    err = C.create_string_buffer(512)
    dig, kind = C.c_uint64(0), C.c_int(-1)
    fn = lib().fqtk_host_fastq_digest
    fn.restype = C.c_int64
    n = fn(str(path).encode(), C.c_uint64(batch), C.c_uint32(helpers), C.byref(dig), C.byref(kind), err, C.c_size_t(512))
    if n < 0:
        raise ValueError(err.value.decode())
    return int(n), int(dig.value), int(kind.value)


def gunzip(data: bytes, cap=None) -> bytes:
    """A gzip file in memory through the streaming decoder of fast_inflate.hpp; ValueError on a corrupt stream."""
    import numpy as np
    cap = (len(data) * 40 + (1 << 20)) if cap is None else cap
    out = np.empty(cap, dtype=np.uint8)
    err = C.create_string_buffer(256)
    fn = lib().fqtk_host_gunzip
    fn.restype = C.c_int64
    n = fn(data, C.c_size_t(len(data)), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), err, C.c_size_t(256))
    if n < 0:
        raise ValueError(err.value.decode())
    assert n <= cap, "test buffer too small"
    return out[:n].tobytes()


def gunzip_parallel(data: bytes, threads=4, cap=None, chunk=0):
    """The same through parallel_gunzip.hpp; returns (bytes, stretches decoded in parallel, stretches that fell back)."""
    import numpy as np
    cap = (len(data) * 40 + (1 << 20)) if cap is None else cap
    out = np.empty(cap, dtype=np.uint8)
    err = C.create_string_buffer(256)
    stats = (C.c_uint64 * 2)()
    fn = lib().fqtk_host_gunzip_parallel
    fn.restype = C.c_int64
    n = fn(data, C.c_size_t(len(data)), out.ctypes.data_as(C.c_void_p), C.c_size_t(cap), C.c_uint(threads), C.c_size_t(chunk), stats, err, C.c_size_t(256))
    if n < 0:
        raise ValueError(err.value.decode())
    assert n <= cap, "test buffer too small"
    return out[:n].tobytes(), int(stats[0]), int(stats[1])


def bgzf(data: bytes, level=5) -> bytes:
    cap = len(data) + len(data) // 8 + 65536
    out = (C.c_uint8 * cap)()
    n = C.c_size_t()
    rc = lib().fqtk_host_bgzf(data, C.c_size_t(len(data)), level, out, C.c_size_t(cap), C.byref(n))
    assert rc == 0
    return bytes(out[:n.value])


def format_f64(v: float) -> str:
    out = C.create_string_buffer(64)
    assert lib().fqtk_host_format_f64(v, out, 64) == 0
    return out.value.decode()


def metrics(counts):
    S = len(counts) - 1
    c = (C.c_uint64 * (S + 1))(*counts)
    f = (C.c_double * (S + 1))()
    m = (C.c_double * (S + 1))()
    b = (C.c_double * (S + 1))()
    lib().fqtk_host_metrics(c, C.c_size_t(S), f, m, b)
    return list(f), list(m), list(b)


def load_samples(path):
    err = C.create_string_buffer(512)
    n = lib().fqtk_host_load_samples(str(path).encode(), err, C.c_size_t(512))
    if n < 0:
        raise ValueError(err.value.decode())
    return n


# ---- helpers mirroring the reference's test helpers (demux.rs:1018-1076) ----------------------------
def fastq_file(tmp, filename_prefix, read_prefix, records_bases, gz=False):
    lines = []
    for i, bases in enumerate(records_bases):
        lines += [f"@{read_prefix}_{i}", bases, "+", ";" * len(bases)]
    text = "".join(l + "\n" for l in lines)
    path = os.path.join(str(tmp), f"{filename_prefix}.fastq" + (".gz" if gz else ""))
    if gz:
        with gzip.open(path, "wt") as fh:
            fh.write(text)
    else:
        with open(path, "w") as fh:
            fh.write(text)
    return path


def metadata_file(tmp, barcodes):
    path = os.path.join(str(tmp), "metadata.tsv")
    with open(path, "w") as fh:
        fh.write("sample_id\tbarcode\n")
        for i, b in enumerate(barcodes):
            fh.write(f"Sample{i:04}\t{b}\n")
    return path


def read_fastq(path):
    with gzip.open(path, "rt") as fh:
        lines = fh.read().split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    assert len(lines) % 4 == 0
    return [(lines[i][1:], lines[i + 1], lines[i + 3]) for i in range(0, len(lines), 4)]


def run_demux(inputs, read_structures, sample_metadata, output, output_types=("T",), unmatched_prefix="unmatched",
              max_mismatches=1, min_mismatch_delta=2, threads=5, compression_level=5, skip_reasons=(), extra=()):
    cmd = [EXE, "demux", "--inputs", *map(str, inputs), "--read-structures", *read_structures,
           "--sample-metadata", str(sample_metadata), "--output", str(output),
           "--unmatched-prefix", unmatched_prefix, "--max-mismatches", str(max_mismatches),
           "--min-mismatch-delta", str(min_mismatch_delta), "--threads", str(threads),
           "--compression-level", str(compression_level)]
    if output_types:
        cmd += ["--output-types", *output_types]
    if skip_reasons:
        cmd += ["--skip-reasons", *skip_reasons]
    cmd += list(extra)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300)
