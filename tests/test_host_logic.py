"""Host-side mirror of the reference's table preconditions (src/lib/samples.rs tests :151-398) and the
synthetic workload generator (host twin of the device generator)."""
import numpy as np
import pytest

from fqtk_amd import Sample, SampleGroup
from fqtk_amd.samples import DelimFileHeaderError, is_valid_iupac
from fqtk_amd import synth


def test_is_valid_iupac(kat):
    for ch in kat["valid_iupac"]["true"]:
        assert is_valid_iupac(ord(ch))
    for ch in kat["valid_iupac"]["false"]:
        assert not is_valid_iupac(ord(ch))


def test_sample_new_validation():
    s = Sample.new(0, "s1", "GATTACA")
    assert (s.sample_id, s.barcode, s.ordinal) == ("s1", "GATTACA", 0)
    with pytest.raises(ValueError, match="Sample name cannot be empty"):
        Sample.new(0, "", "GATTACA")
    with pytest.raises(ValueError, match="Sample barcode cannot be empty"):
        Sample.new(0, "s", "")
    with pytest.raises(ValueError, match="All sample barcode bases must be one of"):
        Sample.new(0, "s", "GATTANX")
    with pytest.raises(ValueError, match="All sample barcode bases must be one of"):
        Sample.new(0, "s", "gattaca")
    Sample.new(0, "s", "NNn..RYKM")


def test_sample_group_validation():
    a, b = Sample("s1", "GATTACA"), Sample("s2", "CATGCTA")
    g = SampleGroup.from_samples([a, b])
    assert [s.ordinal for s in g.samples] == [0, 1]
    with pytest.raises(ValueError, match="Must provide one or more sample"):
        SampleGroup.from_samples([])
    with pytest.raises(ValueError, match="Each sample name must be unique"):
        SampleGroup.from_samples([a, Sample("s1", "CATGCTA")])
    with pytest.raises(ValueError, match="Each sample barcode must be unique"):
        SampleGroup.from_samples([a, Sample("s2", "GATTACA")])
    with pytest.raises(ValueError, match="All barcodes must have the same length"):
        SampleGroup.from_samples([a, Sample("s2", "CATGCT")])


def test_sample_group_from_file(tmp_path):
    p = tmp_path / "meta.tsv"
    p.write_text("sample_id\tbarcode\nsample1\tGATTACA\nsample2\tCATGCTA\n\n\n")
    g = SampleGroup.from_file(str(p))
    assert [(s.sample_id, s.barcode, s.ordinal) for s in g.samples] == [("sample1", "GATTACA", 0),
                                                                        ("sample2", "CATGCTA", 1)]
    bad = tmp_path / "bad.tsv"
    bad.write_text("sample\tbarcode\nsample1\tGATTACA\n")
    with pytest.raises(DelimFileHeaderError) as ei:
        SampleGroup.from_file(str(bad))
    assert ei.value.expected == "sample_id\tbarcode" and ei.value.found == "sample\tbarcode"
    assert Sample.deserialize_header_line() == "sample_id\tbarcode"
    assert str(Sample("test-sample", "GATTACA", 2)) == "Sample(0002) - { name: test-sample\tbarcode: GATTACA }"


def test_sample_group_from_file_reference_cases(tmp_path):
    """Ports of /root/reference/src/lib/samples.rs tests :162-286 (file loading)."""
    p = tmp_path / "sample_metadata.tsv"
    p.write_text("sample_id,barcode\nsample1,GATTACA\nsample2,CATGCTA\n")          # test_tsv_file_delim_error
    with pytest.raises(DelimFileHeaderError) as ei:
        SampleGroup.from_file(str(p))
    assert (ei.value.expected, ei.value.found) == ("sample_id\tbarcode", "sample_id,barcode")
    p.write_text("sample1\tGATTACA\nsample2\tCATGCTA\n")                           # ..._with_no_header
    with pytest.raises(DelimFileHeaderError) as ei:
        SampleGroup.from_file(str(p))
    assert ei.value.found == "sample1\tGATTACA"
    p.write_text("sample_id\tbarcode\n")                                            # test_reading_header_only_file
    with pytest.raises(ValueError, match="Must provide one or more sample"):
        SampleGroup.from_file(str(p))
    p.write_text("\n")                                                              # test_reading_empty_file
    with pytest.raises(ValueError, match="Must provide one or more sample"):
        SampleGroup.from_file(str(p))
    with pytest.raises(FileNotFoundError):                                          # test_reading_non_existent_file
        SampleGroup.from_file(str(tmp_path / "nope.tsv"))
    Sample.new(0, "s_1_example_name", "GATTANN")                                    # non-ACGT bases allowed
    g = SampleGroup.from_samples([Sample("s1", "GATTACA", 0), Sample("s2", "CATGCTA", 0)])
    assert [s.ordinal for s in g.samples] == [0, 1]                                 # ordinals re-assigned (:200-207)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
def test_synthetic_tables_meet_their_spec(k):
    cfg = synth.CONFIGS[k]
    bcs = synth.make_barcodes(cfg)
    assert len(bcs) == cfg.n_samples and len(set(bcs)) == cfg.n_samples
    assert all(len(b) == cfg.barcode_len for b in bcs)
    SampleGroup.from_samples([Sample(f"S{i}", b) for i, b in enumerate(bcs)])
    arr = np.stack([np.frombuffer(b.encode(), dtype=np.uint8) for b in bcs])
    if not cfg.iupac:
        d = (arr[:, None, :] != arr[None, :, :]).sum(axis=2)
        np.fill_diagonal(d, 99)
        assert d.min() >= 3
    else:
        assert any(ch in b for b in bcs for ch in "MRWSYKVHDBN")


def test_synthetic_reads_are_deterministic_and_range_consistent():
    w = synth.Workload(synth.CONFIGS[3])
    a = w.fill_host(1000, 5000)
    b = w.fill_host(1000, 5000)
    c = w.fill_host(3000, 100)
    assert np.array_equal(a, b)
    assert np.array_equal(a[2000:2100], c)          # read i depends only on (seed, i)
    assert a.shape == (5000, 16)
    frac_n = float((a == ord("N")).mean())
    assert 0.002 < frac_n < 0.01
    w5 = synth.Workload(synth.CONFIGS[5])
    x = w5.fill_host(0, 50000)
    assert x.shape == (50000, 12) and np.all(x[:, 10:] == 0)
    assert (x[:, :10] == ord(".")).sum() > 0 and ((x[:, :10] >= ord("a")) & (x[:, :10] <= ord("t"))).sum() > 0


def test_product_package_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under fqtk_amd/ (Python or native sources) may import,
    link or execute it -- the product path has no CPU compute fallback."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for path in glob.glob(os.path.join(root, "fqtk_amd", "**", "*"), recursive=True):
        if not os.path.isfile(path) or path.endswith((".so", ".pyc")) or os.sep + "bin" + os.sep in path:
            continue
        text = open(path, errors="replace").read()
        for m in re.finditer(r"^.*\boracle\b.*$", text, flags=re.M):
            line = m.group(0).strip()
            if line.startswith(("//", "#", "*")) or "//" in line.split("oracle")[0]:
                continue   # prose in a comment
            offenders.append((os.path.relpath(path, root), line))
    assert offenders == []


def test_chunk_schedule_keeps_slots_exclusive_and_devices_in_order():
    """SURVEY 8e without a GPU: chunk k -> device k mod G, slot (k / G) mod S, at most G * S chunks outstanding, collected in
    order -- under any interleaving of submits and completions no (device, slot) pair is handed a second chunk early."""
    import ctypes as C
    import numpy as np
    from tests import hostlib as H
    fn = H.lib().fqtk_host_chunk_schedule_check
    rng = np.random.default_rng(8)
    for devices in (1, 2, 3, 4, 8):
        for slots in (1, 2, 3):
            for n_chunks in (0, 1, devices * slots, 1000):
                for p in (0.0, 0.3, 0.7, 1.0):
                    order = (rng.random(4 * n_chunks + 8) < p).astype(np.uint8)
                    rc = fn(C.c_uint64(devices), C.c_uint64(slots), C.c_uint64(n_chunks), order.ctypes.data_as(C.c_void_p), C.c_size_t(order.size))
                    assert rc == 0, (devices, slots, n_chunks, p, rc)


def test_every_device_has_its_own_submit_thread_and_chunks_come_back_in_order():
    """`fqtk demux --devices a,b,..` (csrc/host/chunk_dispatch.hpp) over a fake device whose submits and collects take random
    times: every device's chunks are submitted by that device's own thread (never two at once on one device, two devices'
    submits DO overlap), a (device, slot) pair gets its next chunk only after the previous one was collected, the
    collector sees 0, 1, 2, .. -- the order every output file is written in (demux.rs:945-977)."""
    import ctypes as C
    from tests import hostlib as H
    fn = H.lib().fqtk_host_chunk_dispatch_check
    for devices, slots, n, seed in ((1, 3, 200, 1), (2, 3, 400, 2), (3, 2, 300, 3), (8, 3, 600, 4), (4, 1, 100, 5), (2, 3, 1, 6), (3, 3, 0, 7)):
        overlap = C.c_int(0)
        assert fn(C.c_uint64(devices), C.c_uint64(slots), C.c_uint64(n), C.c_uint64(seed), C.byref(overlap)) == 0, (devices, slots, n)
        if devices > 1 and n >= 100:
            assert overlap.value == 1, "the devices' submits never overlapped"


def test_the_run_in_a_child_process_passes_status_and_messages_on(tmp_path):
    """`fqtk demux` runs in a child of the process the user starts (csrc/host/demux.cpp: main, supervise, end_process): the parent returns when
    the child reports that every file is closed -- or, when the run fails, waits for the child and passes its status on.  Without a GPU the run
    fails at the matcher (this library has no CPU path): exit status 1, the message on stderr, the partial outputs removed -- the same with
    FQTK_FOREGROUND=1, where no child is made.  With a GPU both forms succeed and leave the same files."""
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(ROOT, "fqtk_amd", "bin", "fqtk")
    if not os.path.exists(exe):
        pytest.skip("fqtk binary not built")
    meta = tmp_path / "meta.tsv"
    meta.write_text("sample_id\tbarcode\nS0\tACGTACGT\nS1\tTTGCAATG\n")
    fq = tmp_path / "r.fq"
    fq.write_text("@r1 x\nACGTACGTAAAA\n+\nFFFFFFFFFFFF\n@r2 x\nTTGCAATGCCCC\n+\nFFFFFFFFFFFF\n")
    seen = {}
    for name, env in (("child", {}), ("foreground", {"FQTK_FOREGROUND": "1"})):
        out = tmp_path / name
        r = subprocess.run([exe, "demux", "-i", str(fq), "-r", "8B+T", "-s", str(meta), "-o", str(out), "-t", "5"],
                           capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
        seen[name] = (r.returncode, sorted(os.listdir(out)) if out.exists() else [])
        if r.returncode != 0:
            assert r.returncode == 1 and "no HIP device" in r.stderr, r.stderr
            assert not [f for f in seen[name][1] if f.endswith(".fq.gz")], "partial outputs must be removed"
        else:
            assert "demux-metrics.txt" in seen[name][1] and "S0.R1.fq.gz" in seen[name][1]
    assert seen["child"] == seen["foreground"]
    # usage errors come back through the parent as well
    r = subprocess.run([exe, "demux", "--no-such-flag"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "unexpected argument" in r.stderr
    r = subprocess.run([exe, "demux", "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "Usage: fqtk demux" in r.stdout


def test_the_committed_hbm_traffic_is_of_the_current_kernel_sources():
    """bench.py reports roofline.traffic from profiles/pmc_traffic.json only when that file was measured on the kernel sources of this
    tree (their digest is in it); after a change to the matcher's kernels the counter passes are run again (tools/profile_bench.sh) and
    the file committed -- or the line says traffic: null."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    have = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))["cfg3"]
    assert have["kernel_sources_sha1"] == bench.kernel_sources_digest(), "run tools/profile_bench.sh on a GPU box and commit profiles/pmc_traffic.json"
    assert 0.99 < have["traffic_bytes_per_launch"] / (have["reads_per_launch"] * 20) < 1.05   # 16 bytes in + 4 out per read
