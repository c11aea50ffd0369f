"""The host side under sanitizers: `python -m fqtk_amd.build --sanitize=address|thread`
builds libfqtk_host.so (and bin/fqtk.<kind>) with ASan + UBSan / TSan; the host tests that drive the threaded and the
byte-level components -- readers with their producer threads, the cutter's counting assistant, parallel gunzip, BGZF,
record formatting, the DEFLATE phases -- then run against that build in a child interpreter (LD_PRELOAD of the
sanitizer runtime).  A report of any kind fails the child."""
import os
import subprocess
import sys

import pytest

from fqtk_amd import build as fb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(kind):
    name = {"address": "libasan.so", "thread": "libtsan.so"}[kind]
    out = subprocess.run([fb.CXX, f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


QUICK = {   # what the default CPU suite runs (about a minute each); FQTK_SANITIZE_FULL=1: both files whole (ASan 2 min, TSan 7 min)
    "address": (["tests/test_record_format.py", "tests/test_host_components.py",
                 "tests/test_bgzf_deflate.py::test_fastq_blocks_round_trip_and_compress"],
                "not cli and not sanitizer and not every_kind_of_input and not hands_out_the_same and not windows_and_damage "
                "and not longer_than_a_piece and not corrupt_and_truncated and not many_pieces"),
    "thread": (["tests/test_record_format.py::test_a_second_thread_counting_for_the_cutter_changes_nothing",
                "tests/test_record_format.py::test_a_cut_stays_mapped_until_its_copier_hands_it_back",
                "tests/test_host_components.py::test_reader_decodes_a_gzip_input_with_several_threads",
                "tests/test_host_components.py::test_large_inputs_plain_gzip_multimember_and_bgzf_block_parallel_agree",
                "tests/test_host_components.py::test_single_stream_gzip_input_through_the_reader_uses_the_fast_decoder",
                "tests/test_host_logic.py::test_every_device_has_its_own_submit_thread_and_chunks_come_back_in_order"], "not sanitizer"),
}
FULL = (["tests/test_record_format.py", "tests/test_host_components.py", "tests/test_bgzf_deflate.py::test_fastq_blocks_round_trip_and_compress"],
        "not cli and not sanitizer")


@pytest.mark.parametrize("kind", ["address", "thread"])
def test_host_suite_under_sanitizer(kind):
    rt = _runtime(kind)
    if rt is None:
        pytest.skip(f"no {kind} sanitizer runtime next to {fb.CXX}")
    tests, expr = FULL if os.environ.get("FQTK_SANITIZE_FULL") else QUICK[kind]
    fb.build_sanitized(kind)
    shim, _ = fb.sanitized_paths(kind)
    env = dict(os.environ, FQTK_HOST_LIB=shim, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1:exitcode=66:report_signal_unsafe=0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu", "-k", expr] + tests,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout, tail
