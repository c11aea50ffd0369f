"""`fqtk demux` end to end on the GPU box: ports of the reference's Demux.execute() integration tests
(/root/reference/src/bin/commands/demux.rs:1103-2073) driven through the C++ CLI, comparing the
DECOMPRESSED per-sample FASTQs exactly as the reference's tests do (read_fastq/assert_equal,
demux.rs:1069-1093), plus a synthetic multi-chunk run checked against the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import hostlib as H  # noqa: E402

S1 = "AAAAAAAAGATTACAGA"
FOUR = [S1, "CCCCCCCCGATTACAGA", "GGGGGGGGGATTACAGA", "GGGGGGTTGATTACAGA"]


def _ok(r):
    assert r.returncode == 0, r.stderr
    return r


@pytest.fixture(autouse=True, params=["device", "host"])
def output_path(request, monkeypatch):
    """Every test runs twice: with the records formatted and compressed on the device (the default: the GPU record
    pipeline of include/fqtk_demux.h) and on the host (--host-output, the reference's division of labour)."""
    if request.param == "host":
        monkeypatch.setenv("FQTK_HOST_OUTPUT", "1")
    else:
        monkeypatch.delenv("FQTK_HOST_OUTPUT", raising=False)
    return request.param


def test_validate_inputs_can_succeed(tmp_path):   # demux.rs:1104-1135
    inputs = [H.fastq_file(tmp_path, "read1", "ex", ["GATTACA"]), H.fastq_file(tmp_path, "read2", "ex", ["TAGGATTA"]),
              H.fastq_file(tmp_path, "index1", "ex", ["GAT"]), H.fastq_file(tmp_path, "index2", "ex", ["TGGG"])]
    meta = H.metadata_file(tmp_path, ["GATTGGG"])
    _ok(H.run_demux(inputs, ["+T", "+T", "+B", "+B"], meta, tmp_path / "output"))
    assert H.read_fastq(tmp_path / "output" / "Sample0000.R1.fq.gz") == [("ex_0 1:N:0:GAT+TGGG", "GATTACA", ";" * 7)]
    assert H.read_fastq(tmp_path / "output" / "Sample0000.R2.fq.gz") == [("ex_0 2:N:0:GAT+TGGG", "TAGGATTA", ";" * 8)]


def test_demux_fragment_reads(tmp_path):   # demux.rs:1293-1333
    meta = H.metadata_file(tmp_path, FOUR)
    fq = H.fastq_file(tmp_path, "ex", "ex", [S1 + "A" * 100])
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["17B100T"], meta, out))
    assert H.read_fastq(out / "Sample0000.R1.fq.gz") == [("ex_0 1:N:0:" + S1, "A" * 100, ";" * 100)]
    for s in ("Sample0001", "Sample0002", "Sample0003", "unmatched"):
        assert H.read_fastq(out / f"{s}.R1.fq.gz") == []
    lines = open(out / "demux-metrics.txt").read().splitlines()
    assert lines[0] == "sample_id\tbarcode\ttemplates\tfrac_templates\tratio_to_mean\tratio_to_best"
    assert lines[1].split("\t")[:4] == ["Sample0000", S1, "1", "1.0"]
    assert lines[-1].split("\t")[:3] == ["unmatched", ".", "0"]


def test_output_type_reads(tmp_path):   # demux.rs:1336-1418
    meta = H.metadata_file(tmp_path, ["AAAAAAAA", "CCCCCCCC", "GGGGGGGG", "TTTTTTTT"])
    fq = H.fastq_file(tmp_path, "ex", "ex", ["ATCGATCGAT" + "AAAAAAAA" + "GATTACA" + "A" * 100])
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["10M8B7C100T"], meta, out, output_types=["T", "B", "M", "C"]))
    head = "ex_0:ATCGATCGAT 1:N:0:AAAAAAAA"
    assert H.read_fastq(out / "Sample0000.R1.fq.gz") == [(head, "A" * 100, ";" * 100)]
    assert H.read_fastq(out / "Sample0000.I1.fq.gz") == [(head, "AAAAAAAA", ";" * 8)]
    assert H.read_fastq(out / "Sample0000.U1.fq.gz") == [(head, "ATCGATCGAT", ";" * 10)]
    assert H.read_fastq(out / "Sample0000.C1.fq.gz") == [(head, "GATTACA", ";" * 7)]


def test_demux_with_catchall_barcode(tmp_path):   # demux.rs:1421-1462
    meta = H.metadata_file(tmp_path, ["NNNNNNN"])
    fq = H.fastq_file(tmp_path, "ex", "ex", ["NNNNNNN" + "A" * 100])
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["7B+T"], meta, out, max_mismatches=0))
    assert H.read_fastq(out / "unmatched.R1.fq.gz") == []
    assert H.read_fastq(out / "Sample0000.R1.fq.gz") == [("ex_0 1:N:0:NNNNNNN", "A" * 100, ";" * 100)]


def test_demux_with_iupac_bases_in_barcode(tmp_path):   # demux.rs:1465-1538
    meta = H.metadata_file(tmp_path, ["MMMMMMM", "KKKKKKK"])
    reads = ["AAAAAAA" + "A" * 5, "CCCCCCC" + "A" * 5, "ACACACA" + "A" * 5, "GTGTGTG" + "C" * 5, "TGTGTGT" + "C" * 5,
             "CGCGCGC" + "T" * 5]
    fq = H.fastq_file(tmp_path, "ex", "ex", reads)
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["7B+T"], meta, out, max_mismatches=0, min_mismatch_delta=0))
    s0 = H.read_fastq(out / "Sample0000.R1.fq.gz")
    assert [h for h, _, _ in s0] == ["ex_0 1:N:0:AAAAAAA", "ex_1 1:N:0:CCCCCCC", "ex_2 1:N:0:ACACACA"]
    s1 = H.read_fastq(out / "Sample0001.R1.fq.gz")
    assert s1 == [("ex_3 1:N:0:GTGTGTG", "C" * 5, ";" * 5), ("ex_4 1:N:0:TGTGTGT", "C" * 5, ";" * 5)]
    assert H.read_fastq(out / "unmatched.R1.fq.gz") == [("ex_5 1:N:0:CGCGCGC", "T" * 5, ";" * 5)]


def test_demux_with_ns_in_barcode(tmp_path):   # demux.rs:1541-1611
    meta = H.metadata_file(tmp_path, ["NNAAAAA", "NNCCCCC"])
    fq = H.fastq_file(tmp_path, "ex", "ex", ["ANAAAAA" + "A" * 5, "ANCCCCC" + "C" * 5, "NNNAAAA" + "T" * 5])
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["7B+T"], meta, out, max_mismatches=0, min_mismatch_delta=0))
    assert H.read_fastq(out / "Sample0000.R1.fq.gz") == [("ex_0 1:N:0:ANAAAAA", "A" * 5, ";" * 5)]
    assert H.read_fastq(out / "Sample0001.R1.fq.gz") == [("ex_1 1:N:0:ANCCCCC", "C" * 5, ";" * 5)]
    assert H.read_fastq(out / "unmatched.R1.fq.gz") == [("ex_2 1:N:0:NNNAAAA", "T" * 5, ";" * 5)]


def test_demux_paired_reads_with_in_line_sample_barcodes(tmp_path):   # demux.rs:1614-1672
    meta = H.metadata_file(tmp_path, FOUR)
    r1 = H.fastq_file(tmp_path, "ex_R1", "ex", [S1[:8] + "A" * 100])
    r2 = H.fastq_file(tmp_path, "ex_R2", "ex", [S1[8:] + "T" * 100])
    out = tmp_path / "output"
    _ok(H.run_demux([r1, r2], ["8B100T", "9B100T"], meta, out))
    assert H.read_fastq(out / "Sample0000.R1.fq.gz") == [("ex_0 1:N:0:AAAAAAAA+GATTACAGA", "A" * 100, ";" * 100)]
    assert H.read_fastq(out / "Sample0000.R2.fq.gz") == [("ex_0 2:N:0:AAAAAAAA+GATTACAGA", "T" * 100, ";" * 100)]


def test_demux_dual_indexed_paired_end_reads(tmp_path):   # demux.rs:1675-1736
    meta = H.metadata_file(tmp_path, FOUR)
    ins = [H.fastq_file(tmp_path, "ex_I1", "ex", [S1[:8]]), H.fastq_file(tmp_path, "ex_R1", "ex", ["A" * 100]),
           H.fastq_file(tmp_path, "ex_R2", "ex", ["T" * 100]), H.fastq_file(tmp_path, "ex_I2", "ex", [S1[8:]])]
    out = tmp_path / "output"
    _ok(H.run_demux(ins, ["8B", "100T", "100T", "9B"], meta, out))
    assert H.read_fastq(out / "Sample0000.R1.fq.gz") == [("ex_0 1:N:0:AAAAAAAA+GATTACAGA", "A" * 100, ";" * 100)]
    assert H.read_fastq(out / "Sample0000.R2.fq.gz") == [("ex_0 2:N:0:AAAAAAAA+GATTACAGA", "T" * 100, ";" * 100)]


def test_demux_a_weird_set_of_reads(tmp_path):   # demux.rs:1739-1800
    meta = H.metadata_file(tmp_path, FOUR)
    ins = [H.fastq_file(tmp_path, "example_1", "ex", ["AAAACCCCGGGGTTTT"]), H.fastq_file(tmp_path, "example_2", "ex", ["A" * 104]),
           H.fastq_file(tmp_path, "example_3", "ex", ["T" * 100 + "GAT"]), H.fastq_file(tmp_path, "example_4", "ex", ["TACAGAAAT"])]
    out = tmp_path / "output"
    _ok(H.run_demux(ins, ["4B4M8S", "4B100T", "100S3B", "6B1S1M1T"], meta, out))
    head = "ex_0:CCCC+A 1:N:0:AAAA+AAAA+GAT+TACAGA"
    assert H.read_fastq(out / "Sample0000.R1.fq.gz") == [(head, "A" * 100, ";" * 100)]
    assert H.read_fastq(out / "Sample0000.R2.fq.gz") == [(head.replace(" 1:", " 2:"), "T", ";")]


def test_demux_multiple_templates_in_one_read(tmp_path):   # demux.rs:1803-1876
    meta = H.metadata_file(tmp_path, FOUR)
    fq = H.fastq_file(tmp_path, "ex", "ex", [S1 + "A" * 20 + "C" * 20 + "T" * 20 + "C" * 20 + "G" * 20])
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["17B20T20S20T20S20T"], meta, out))
    for n, base in ((1, "A"), (2, "T"), (3, "G")):
        assert H.read_fastq(out / f"Sample0000.R{n}.fq.gz") == [(f"ex_0 {n}:N:0:" + S1, base * 20, ";" * 20)]


def _short_read_inputs(tmp_path):
    bc = "GATTGGG"
    r1 = H.fastq_file(tmp_path, "read1", "ex", ["AAAAAAA", "CCCCCCC", ""])
    i1 = H.fastq_file(tmp_path, "index1", "ex", [bc, bc, bc])
    return r1, i1, H.metadata_file(tmp_path, [bc])


def test_fails_if_reads_too_short(tmp_path):   # demux.rs:1984-2020
    r1, i1, meta = _short_read_inputs(tmp_path)
    r = H.run_demux([r1, i1], ["+T", "7B"], meta, tmp_path / "output", output_types=["T", "B"])
    assert r.returncode != 0
    assert "Read ex_2 had too few bases to demux 0 vs. 1 needed in read structure +T." in r.stderr


def test_skip_reads_too_short(tmp_path):   # demux.rs:2023-2073
    r1, i1, meta = _short_read_inputs(tmp_path)
    out = tmp_path / "output"
    _ok(H.run_demux([r1, i1], ["+T", "7B"], meta, out, output_types=["T", "B"], skip_reasons=["too-few-bases"]))
    rows = [l.split("\t") for l in open(out / "demux-metrics.txt").read().splitlines()[1:]]
    assert sum(int(r[2]) for r in rows) == 2
    assert [r for r in rows if r[0] == "Sample0000"][0][2] == "2"
    assert len(H.read_fastq(out / "Sample0000.R1.fq.gz")) == 2
    assert len(H.read_fastq(out / "Sample0000.I1.fq.gz")) == 2


def test_sources_out_of_sync_is_fatal(tmp_path):
    r1 = H.fastq_file(tmp_path, "read1", "ex", ["AAAAAAA", "CCCCCCC"])
    i1 = H.fastq_file(tmp_path, "index1", "ex", ["GATTGGG"])
    r = H.run_demux([r1, i1], ["+T", "7B"], H.metadata_file(tmp_path, ["GATTGGG"]), tmp_path / "output")
    assert r.returncode != 0 and "FASTQ sources out of sync" in r.stderr


def test_overlong_barcode_is_fatal_like_the_reference_panic(tmp_path):
    fq = H.fastq_file(tmp_path, "ex", "ex", ["ACGTACGTAA" + "A" * 10])
    r = H.run_demux([fq], ["10B+T"], H.metadata_file(tmp_path, ["ACGTACGT", "TTTTACGT"]), tmp_path / "output")
    assert r.returncode != 0 and "differs from expected barcode (" in r.stderr
    # the reference's panic sentence, naming sample 0 by its id (barcode_matching.rs:95-107) ...
    assert "Read barcode (ACGTACGTAA) length (10) differs from expected barcode (ACGTACGT) length (8) for sample Sample0000" in r.stderr
    # ... and a fatal error leaves no truncated .fq.gz behind (a BGZF file without its EOF block looks complete)
    assert not list((tmp_path / "output").glob("*.fq.gz")), "partially written outputs must be removed on a fatal error"


def test_late_fatal_error_removes_every_partial_output(tmp_path):
    """20 000 good templates, then one whose barcode read is too long: blocks have already been written to
    the per-sample files when the error is found; none of them may survive."""
    r1 = H.fastq_file(tmp_path, "r1", "ex", ["A" * 60] * 20001)
    i1 = H.fastq_file(tmp_path, "i1", "ex", ["ACGTACGT"] * 20000 + ["ACGTACGTAA"])
    r = H.run_demux([r1, i1], ["+T", "+B"], H.metadata_file(tmp_path, ["ACGTACGT", "TTTTACGT"]), tmp_path / "output",
                    extra=["--chunk-reads", "5000"])
    assert r.returncode != 0 and "differs from expected barcode (" in r.stderr
    assert "removed" in r.stderr and not list((tmp_path / "output").glob("*.fq.gz"))


def test_chunk_round_robin_over_devices_keeps_order_and_counts(tmp_path, output_path):
    """SURVEY 8e in the CLI: chunk k is matched on devices[k mod G] (table replicated, counts summed).
    The GPU box has one device, so the list repeats it -- the routing logic is what is under test:
    outputs must be byte-identical (after decompression) to the single-device run."""
    from fqtk_amd import synth
    cfg = synth.CONFIGS[2]
    w = synth.Workload(cfg)
    n = 20_000
    bcs = w.fill_host(0, n)
    reads = [bytes(bcs[i]).decode() + "ACGTACGTAC" for i in range(n)]
    fq = H.fastq_file(tmp_path, "r", "q", reads)
    meta = os.path.join(str(tmp_path), "metadata.tsv")
    with open(meta, "w") as fh:
        fh.write("sample_id\tbarcode\n" + "".join(f"S{i}\t{b}\n" for i, b in enumerate(w.barcodes)))
    outs = []
    runs = [("one", ["--chunk-reads", "1500"]), ("three", ["--chunk-reads", "1500", "--devices", "0,0,0"])]
    import ctypes as C
    from fqtk_amd import _lib
    ndev = C.c_int(0)
    _lib.load().fqtk_device_count(C.byref(ndev))
    if ndev.value > 1:   # real distinct devices: the counts then come back through the RCCL all-reduce
        runs.append(("distinct", ["--chunk-reads", "1500", "--devices", ",".join(str(d) for d in range(min(ndev.value, 4)))]))
    for tag, extra in runs:
        out = tmp_path / tag
        r = _ok(H.run_demux([fq], ["8B10T"], meta, out, threads=6, extra=extra))
        if tag == "distinct" and output_path == "host":   # (the record pipeline's counts are summed on the host)
            assert "all-reduced" in r.stderr
        outs.append(out)
    names = [f"S{i}" for i in range(cfg.n_samples)] + ["unmatched"]
    total = 0
    for name in names:
        a = H.read_fastq(outs[0] / f"{name}.R1.fq.gz")
        for other in outs[1:]:
            assert a == H.read_fastq(other / f"{name}.R1.fq.gz")
        total += len(a)
    assert total == n
    for other in outs[1:]:
        assert open(outs[0] / "demux-metrics.txt").read() == open(other / "demux-metrics.txt").read()


def test_synthetic_multi_chunk_gz_inputs_match_the_oracle(tmp_path):
    """cfg 4 shape (R1 16C8B126T + R2 150T, outputs T and C), several GPU chunks, gz inputs, all worker
    threads: per-sample record lists must equal the ones the CPU oracle's assignments imply, in input
    order, and demux-metrics.txt must carry the oracle's counts."""
    from fqtk_amd import synth
    from oracle import oracle as O
    cfg = synth.CONFIGS[4]
    w = synth.Workload(cfg)
    n = 30_000
    bcs = w.fill_host(0, n)[:, :8]
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    cell = acgt[rng.integers(0, 4, size=(n, 16))]
    t1 = acgt[rng.integers(0, 4, size=(n, 30))]
    t2 = acgt[rng.integers(0, 4, size=(n, 40))]
    r1 = [bytes(cell[i]).decode() + bytes(bcs[i]).decode() + bytes(t1[i]).decode() for i in range(n)]
    r2 = [bytes(t2[i]).decode() for i in range(n)]
    f1 = H.fastq_file(tmp_path, "r1", "q", r1, gz=True)
    f2 = H.fastq_file(tmp_path, "r2", "q", r2, gz=True)
    meta = os.path.join(str(tmp_path), "metadata.tsv")
    with open(meta, "w") as fh:
        fh.write("sample_id\tbarcode\n" + "".join(f"S{i}\t{b}\n" for i, b in enumerate(w.barcodes)))
    out = tmp_path / "output"
    _ok(H.run_demux([f1, f2], ["16C8B30T", "40T"], meta, out, output_types=["T", "C"], threads=8,
                    extra=["--chunk-reads", "7000"]))
    lit = O.RefLiteral(w.barcodes, 1, 2, True)
    idx, _, _, counts = lit.assign_batch(np.ascontiguousarray(bcs))
    names = [f"S{i}" for i in range(cfg.n_samples)] + ["unmatched"]
    for s, name in enumerate(names):
        sel = np.nonzero(idx == (0xFFFF if s == cfg.n_samples else s))[0]
        exp_r1 = [(f"q_{i} 1:N:0:" + bytes(bcs[i]).decode(), bytes(t1[i]).decode(), ";" * 30) for i in sel]
        exp_r2 = [(f"q_{i} 2:N:0:" + bytes(bcs[i]).decode(), bytes(t2[i]).decode(), ";" * 40) for i in sel]
        exp_c1 = [(f"q_{i} 1:N:0:" + bytes(bcs[i]).decode(), bytes(cell[i]).decode(), ";" * 16) for i in sel]
        assert H.read_fastq(out / f"{name}.R1.fq.gz") == exp_r1
        assert H.read_fastq(out / f"{name}.R2.fq.gz") == exp_r2
        assert H.read_fastq(out / f"{name}.C1.fq.gz") == exp_c1
    rows = [l.split("\t") for l in open(out / "demux-metrics.txt").read().splitlines()[1:]]
    assert [r[0] for r in rows] == names
    assert [int(r[2]) for r in rows] == [int(c) for c in counts]


def test_plate_of_384_samples_with_12_plus_12_dual_indexes(tmp_path):
    """I1 12B + R1 40T + I2 12B, 384 samples: the matcher inside the record pipeline serves this plate from the LDS form of
    three-byte entries behind a perfect hash (round 6; the HBM/L2 table before), and index reads with ambiguity codes, '.' and
    bytes of no meaning go through its first-pass recode / second pass.  Per-sample records in input order and the metrics file
    against the oracle's assignments."""
    from oracle import oracle as O
    rng = np.random.default_rng(1212)
    seen = set()
    while len(seen) < 384:
        seen.add("".join(rng.choice(list("ACGT"), size=24)))
    barcodes = sorted(seen)
    n = 40_000
    bc = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in barcodes])[rng.integers(0, 384, n)].copy()
    flip = rng.random(bc.shape) < 0.02
    bc[flip] = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, int(flip.sum()))]
    odd = rng.random(bc.shape) < 0.004
    bc[odd] = np.frombuffer(b"RYKMSWBDHVU.ryn", dtype=np.uint8)[rng.integers(0, 15, int(odd.sum()))]
    bc[rng.random(n) < 0.05] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 24)]
    t1 = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, 40))]
    i1 = [bytes(bc[i, :12]).decode() for i in range(n)]
    i2 = [bytes(bc[i, 12:]).decode() for i in range(n)]
    r1 = [bytes(t1[i]).decode() for i in range(n)]
    ins = [H.fastq_file(tmp_path, "i1", "q", i1, gz=True), H.fastq_file(tmp_path, "r1", "q", r1, gz=True), H.fastq_file(tmp_path, "i2", "q", i2, gz=True)]
    meta = os.path.join(str(tmp_path), "metadata.tsv")
    with open(meta, "w") as fh:
        fh.write("sample_id\tbarcode\n" + "".join(f"S{i:03}\t{b}\n" for i, b in enumerate(barcodes)))
    out = tmp_path / "output"
    _ok(H.run_demux(ins, ["12B", "40T", "12B"], meta, out, threads=8, extra=["--chunk-reads", "9000"]))
    idx, _, _, counts = O.RefLiteral(barcodes, 1, 2, True).assign_batch(np.ascontiguousarray(bc))
    rows = [l.split("\t") for l in open(out / "demux-metrics.txt").read().splitlines()[1:]]
    assert len(rows) == 385 and [int(r[2]) for r in rows] == [int(c) for c in counts]
    assert 0.5 < 1 - counts[-1] / n < 0.97
    for s in list(np.unique(idx[idx != 0xFFFF])[:20]) + [0xFFFF]:
        name = "unmatched" if s == 0xFFFF else f"S{int(s):03}"
        sel = np.nonzero(idx == s)[0]
        exp = [(f"q_{i} 1:N:0:{i1[i]}+{i2[i]}", r1[i], ";" * 40) for i in sel]
        assert H.read_fastq(out / f"{name}.R1.fq.gz") == exp


def test_cfg5_shape_1536_iupac_samples_inline_barcode_plus_template(tmp_path):
    """cfg 5 shape end to end: 1536 IUPAC-degenerate samples (1537 output files: the CLI must raise its
    fd limit), `10B+T` with variable-length templates, counts and routing checked against the oracle."""
    from fqtk_amd import synth
    from oracle import oracle as O
    cfg = synth.CONFIGS[5]
    w = synth.Workload(cfg)
    n = 6000
    bcs = w.fill_host(0, n)[:, :10]
    rng = np.random.default_rng(8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = [bytes(bcs[i]).decode() + bytes(acgt[rng.integers(0, 4, size=int(rng.integers(1, 40)))]).decode() for i in range(n)]
    fq = H.fastq_file(tmp_path, "r", "q", reads, gz=True)
    meta = os.path.join(str(tmp_path), "metadata.tsv")
    with open(meta, "w") as fh:
        fh.write("sample_id\tbarcode\n" + "".join(f"S{i:04}\t{b}\n" for i, b in enumerate(w.barcodes)))
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["10B+T"], meta, out, threads=8))
    lit = O.RefLiteral(w.barcodes, 1, 2, True)
    idx, _, _, counts = lit.assign_batch(np.ascontiguousarray(bcs))
    rows = [l.split("\t") for l in open(out / "demux-metrics.txt").read().splitlines()[1:]]
    assert len(rows) == 1537 and [int(r[2]) for r in rows] == [int(c) for c in counts]
    for s in list(np.unique(idx[idx != 0xFFFF])[:25]) + [0xFFFF]:
        name = "unmatched" if s == 0xFFFF else f"S{int(s):04}"
        sel = np.nonzero(idx == s)[0]
        exp = [(f"q_{i} 1:N:0:" + bytes(bcs[i]).decode(), reads[i][10:], ";" * (len(reads[i]) - 10)) for i in sel]
        assert H.read_fastq(out / f"{name}.R1.fq.gz") == exp


def test_device_and_host_outputs_are_valid_bgzf_with_identical_content(tmp_path, monkeypatch):
    """By default the records are formatted and DEFLATE-compressed on the device (include/fqtk_demux.h), with
    --host-output by the host's router / libdeflate threads.  Same records in the same order in every file (the reference's tests compare decompressed
    content, demux.rs:1069-1093), same metrics, and every file is well-formed BGZF: members with the BC field, BSIZE,
    CRC32 and ISIZE that check out, and the EOF marker."""
    import gzip
    import struct
    import zlib
    from fqtk_amd import synth
    cfg = synth.CONFIGS[2]
    w = synth.Workload(cfg)
    n = 60_000
    bcs = w.fill_host(0, n)
    rng = __import__("numpy").random.default_rng(3)
    tails = __import__("numpy").frombuffer(b"ACGT", dtype="uint8")[rng.integers(0, 4, size=(n, 60))]
    reads = [bytes(bcs[i]).decode() + bytes(tails[i]).decode() for i in range(n)]
    fq = H.fastq_file(tmp_path, "r", "q", reads)
    meta = os.path.join(str(tmp_path), "metadata.tsv")
    with open(meta, "w") as fh:
        fh.write("sample_id\tbarcode\n" + "".join(f"S{i}\t{b}\n" for i, b in enumerate(w.barcodes)))
    outs = []
    monkeypatch.delenv("FQTK_HOST_OUTPUT", raising=False)
    for tag, extra in (("cpu", ["--host-output"]), ("gpu", ["--gpu-bgzf"])):
        out = tmp_path / tag
        r = _ok(H.run_demux([fq], ["8B60T"], meta, out, threads=8, extra=extra + ["--chunk-reads", "7000"]))
        assert ("GPU record pipeline:" in r.stderr) == (tag == "gpu")
        outs.append(out)
    names = [f"S{i}" for i in range(cfg.n_samples)] + ["unmatched"]
    total = 0
    for name in names:
        a = H.read_fastq(outs[0] / f"{name}.R1.fq.gz")
        assert a == H.read_fastq(outs[1] / f"{name}.R1.fq.gz")
        total += len(a)
        raw = open(outs[1] / f"{name}.R1.fq.gz", "rb").read()
        assert raw.endswith(bytes([0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
        off, text = 0, b""
        while off < len(raw):                                   # walk the members
            assert raw[off:off + 4] == b"\x1f\x8b\x08\x04" and raw[off + 12:off + 16] == b"BC\x02\x00"
            bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
            payload = raw[off + 18:off + bsize - 8]
            crc, isize = struct.unpack_from("<II", raw, off + bsize - 8)
            data = zlib.decompress(payload, -15)
            assert len(data) == isize and zlib.crc32(data) == crc and isize <= 65280
            text += data
            off += bsize
        assert text == gzip.decompress(raw)
    assert total == n
    assert open(outs[0] / "demux-metrics.txt").read() == open(outs[1] / "demux-metrics.txt").read()


def test_compression_level_zero_writes_stored_blocks(tmp_path, output_path):
    """--compression-level 0 (demux.rs:641-643 hands the level to libdeflate, whose level 0 stores): valid BGZF whose
    members are stored DEFLATE blocks, on the device as on the host."""
    import struct
    meta = H.metadata_file(tmp_path, FOUR)
    reads = [S1 + "ACGT" * 25] * 3000
    fq = H.fastq_file(tmp_path, "ex", "ex", reads)
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["17B100T"], meta, out, compression_level=0))
    got = H.read_fastq(out / "Sample0000.R1.fq.gz")
    assert got == [(f"ex_{i} 1:N:0:" + S1, "ACGT" * 25, ";" * 100) for i in range(3000)]
    raw = open(out / "Sample0000.R1.fq.gz", "rb").read()
    plain = sum(len(h) + 1 + 100 + 3 + 100 + 2 for h, _, _ in got)
    assert plain < len(raw) < plain + 40 * (plain // 65280 + 2)
    assert raw[18] & 7 == 1                                          # BFINAL = 1, BTYPE = 00: stored


@pytest.mark.parametrize("level", [1, 9])
def test_other_compression_levels_give_the_same_records(tmp_path, output_path, level):
    """--compression-level (demux.rs:641-643): any level, the same records; on the device 1-3 parse faster, 4+ as the default."""
    meta = H.metadata_file(tmp_path, FOUR)
    reads = [FOUR[i % 4] + "ACGT" * 25 for i in range(4000)]
    fq = H.fastq_file(tmp_path, "ex", "ex", reads)
    out = tmp_path / "output"
    _ok(H.run_demux([fq], ["17B100T"], meta, out, compression_level=level))
    for s in range(4):
        assert H.read_fastq(out / f"Sample000{s}.R1.fq.gz") == [(f"ex_{i} 1:N:0:" + FOUR[s], "ACGT" * 25, ";" * 100) for i in range(s, 4000, 4)]


def test_two_hundred_million_templates_through_771_files(tmp_path, output_path):
    """The pipeline at the size of a real run: cfg 3's shape, 192 M templates (the first 1 M repeated; 149 GB of plain
    FASTQ on RAM-backed scratch), 771 output files.  The metrics file must carry 192 x the first block's per-sample
    counts (oracle) and the host footprint stays that of a few chunks in flight."""
    import shutil
    import sys
    if output_path == "host":
        pytest.skip("the device output path is what scales to this size in a test's time budget")
    need = 200 << 30
    if shutil.disk_usage("/dev/shm").free < need or (os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")) < 2 * need:
        pytest.skip("needs 200 GB of RAM-backed scratch")
    try:
        if int(open("/sys/fs/cgroup/memory.max").read()) < need + (40 << 30):
            pytest.skip("memory cgroup too small for 150 GB of inputs on /dev/shm")
    except (OSError, ValueError):
        pass
    sys.path.insert(0, os.path.join(H.ROOT, "tools"))
    import scope_bench
    from fqtk_amd import synth
    from oracle import oracle as O
    n = 192_000_000
    tmp = scope_bench.scratch_dir(n * 1000)
    try:
        cfg = synth.CONFIGS[3]
        w = synth.Workload(cfg)
        lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
        bcs = w.fill_host(0, 1_000_000)
        idx, _, _, counts = lit.assign_batch(bcs)
        e = scope_bench.scope_e(n, 16, False, tmp, counts * np.uint64(192), repeat_first_block=True, out_name="out_keep")
        assert e["output_files"] == 771 and e["peak_rss_MB"] < 8000, e
        print("192 M templates:", {k: e[k] for k in ("seconds", "M_templates_per_s", "M_templates_per_s_steady", "output_MB", "peak_rss_MB")})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def test_four_hundred_million_templates_from_bgzf_inputs(tmp_path, output_path):
    """north_star's literal size, files -> files: cfg 3's shape, 400 M templates (R1 / I1 / I2 / R2, the first 1 M records
    repeated) as BGZF inputs inflated on the device, 771 output files.  The metrics file must carry 400 x the first block's
    per-sample counts (oracle).  (As plain text the inputs would be 310 GB; BGZF they fit RAM-backed scratch.)"""
    import shutil
    import sys
    if output_path == "host":
        pytest.skip("the device output path is what scales to this size in a test's time budget")
    need = 140 << 30
    if shutil.disk_usage("/dev/shm").free < need or (os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")) < 2 * need:
        pytest.skip("needs 140 GB of RAM-backed scratch")
    try:
        if int(open("/sys/fs/cgroup/memory.max").read()) < need + (40 << 30):
            pytest.skip("memory cgroup too small for the inputs and outputs on /dev/shm")
    except (OSError, ValueError):
        pass
    sys.path.insert(0, os.path.join(H.ROOT, "tools"))
    import scope_bench
    from fqtk_amd import synth
    from oracle import oracle as O
    reps = 400
    tmp = scope_bench.scratch_dir(need)
    try:
        cfg = synth.CONFIGS[3]
        w = synth.Workload(cfg)
        lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
        idx, _, _, counts = lit.assign_batch(w.fill_host(0, 1_000_000))
        paths, meta, _ = scope_bench.make_inputs(tmp, 1_000_000, False)
        bgz = scope_bench.bgzf_repeated(paths, reps=reps)
        for p in paths:
            os.unlink(p)
        e = scope_bench.scope_e(reps * 1_000_000, 16, "bgzf", tmp, counts * np.uint64(reps), inputs=(bgz, meta), out_name="out_keep")
        assert e["output_files"] == 771 and e["peak_rss_MB"] < 12000, e
        assert any("inflated on the device" in t for t in e["timeline"])
        print("400 M templates:", {k: e[k] for k in ("seconds", "M_templates_per_s", "M_templates_per_s_steady", "input_MB", "output_MB", "peak_rss_MB")})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
