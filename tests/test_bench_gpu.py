"""bench.py on the GPU box: the scopes / CPU rows / parity gate of the JSON line, the RCCL code path with
one rank, and the --gpus N self-launch (N real ranks when the box has them, a loud refusal when not)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=900):
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    return r


def _line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_bench_line_carries_scopes_create_time_cpu_rows_and_the_full_parity_gate():
    r = _run(["--reads", "3000000", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1", "--e2e-templates", "4000000", "--e2e-threads", "8"])
    line = r.stdout.strip().splitlines()[-1]
    d = _line(r)
    # the line fits the tail its consumer keeps, and carries every field of the contract and every scope as numbers
    assert len(line) < 2000, len(line)
    assert d["n_gpus"] == 1 and d["unit"] == "M reads/s" and d["value"] > 0 and d["higher_is_better"] is True and d["dtype"] == "u8"
    assert d["scope"] == "K"   # what `value` is, next to it
    assert "bit-exact" in d["config"]["parity"] and "count vector of all 3000000 reads" in d["config"]["parity"]
    assert d["config"]["workload"].startswith("cfg3")
    assert set(d["scopes"]) == {"B", "B_packed", "bgzf_kernel_GBps", "inflate_kernel_GBps", "E", "E_host", "E_gz", "E_gz_host", "E_bgzf", "is"}
    for k, n in (("E", 4), ("E_host", 1), ("E_gz", 4), ("E_gz_host", 4), ("E_bgzf", 4)):   # the compressed-input rows at row E's size
        wall, steady, m = d["scopes"][k]
        assert m == n and wall > 0 and (steady is None or steady > 0)
    assert d["scopes"]["bgzf_kernel_GBps"] > 5 and d["scopes"]["inflate_kernel_GBps"] > 5 and d["scopes"]["B"] > 0 and d["scopes"]["B_packed"] > 0
    assert d["create_ms"] > 0 and "gpu_over_cpu" not in json.dumps(d)
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0 and cb["cache_off"] > 0 and cb["all_cores"][1] >= 1 and "oracle/ref_literal.c" in cb["sample"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert rf["kernel_ms"] <= d["ms_per_step"] * 1.05
    # the whole record is in the side file the line names
    full = json.load(open(os.path.join(ROOT, d["detail"])))
    assert full["value"] == d["value"] and full["roofline"]["frac"] == rf["frac"]
    assert set(full["scopes"]) == {"K", "B", "B_packed", "bgzf_kernel", "inflate_kernel", "E", "E_host", "E_gz", "E_gz_host", "E_bgzf"}
    sc = full["scopes"]
    assert sc["bgzf_kernel"]["hbm"]["GB_per_s_in"] > 5 and 0.2 < sc["bgzf_kernel"]["hbm"]["ratio"] < 0.5
    assert sc["B_packed"]["M_reads_per_s"] > 0 and sc["B_packed"]["packed_bytes_per_read"] == 8
    for k, n in (("E", 4000000), ("E_host", 1000000), ("E_gz", 4000000), ("E_gz_host", 4000000), ("E_bgzf", 4000000)):   # device output (default), host output, gzip inputs, BGZF inputs inflated on the device
        assert sc[k]["templates"] == n and sc[k]["metrics_vs_oracle"] == "per-sample counts identical"
        assert sc[k]["peak_rss_MB"] > 0 and sc[k]["output_files"] == 771
    assert sc["E"]["M_templates_per_s_steady"] > 0 and sc["E_gz"]["M_templates_per_s_steady"] > 0
    assert sc["E_host"]["extra_args"] == ["--host-output"] and sc["E_gz"]["gz_inputs"]
    assert sc["B"]["M_reads_per_s"] > 0 and sc["B"]["GB_per_s_over_pcie"] > 0
    assert any("inflated on the device" in t for t in sc["E_bgzf"]["timeline"]) and sc["inflate_kernel"]["text_GBps"] > 5
    assert full["cpu_baseline"]["cache_off_1core"]["value"] > 0 and full["cpu_baseline"]["all_cores"]["cores"] >= 1


@pytest.mark.gpu
def test_rccl_path_with_one_rank_allreduced_counts_match_the_oracle():
    d = _line(_run(["--reads", "4000000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-scopes",
                    "--parity", "full"], env={"FQTK_BENCH_FORCE_DIST": "1"}))
    assert d["n_gpus"] == 1 and "count vector of all 4000000 reads" in d["config"]["parity"]


def _check_device_scopes(d, devices):
    """The N-device line carries all three scopes: K = value, B through one matcher per device, E through ONE
    `fqtk demux --devices ..` process from plain and from BGZF inputs (inflated on the devices), counts = the oracle's."""
    sc = d["scopes"]
    assert sc["devices"] == devices and sc["B"] > 0
    for k in ("E", "E_bgzf"):
        wall, steady, m = sc[k]
        assert m == 4 and wall > 0 and steady > 0
    full = json.load(open(os.path.join(ROOT, d["detail"])))["scopes"]
    assert full["B"]["devices"] == [int(x) for x in devices.split(",")]
    for k in ("E", "E_bgzf"):
        assert full[k]["extra_args"] == ["--devices", devices] and full[k]["metrics_vs_oracle"] == "per-sample counts identical"
    assert any("inflated on the device" in t for t in full["E_bgzf"]["timeline"])
    assert not any("host's reader threads" in t for t in full["E_bgzf"]["timeline"])


@pytest.mark.gpu
def test_gpus_n_launches_n_ranks_itself_or_refuses():
    import torch
    have = torch.cuda.device_count()
    if have >= 2:
        for scaling in ("weak", "strong"):
            d = _line(_run(["--gpus", "2", "--reads", "4000000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0",
                            "--scaling", scaling, "--parity", "full", "--e2e-templates", "4000000", "--e2e-threads", "8"]))
            assert d["n_gpus"] == 2 and d["scaling"] == scaling and len(d["roofline"]["kernel_ms_per_rank"]) == 2
            assert d["config"]["reads_per_step_whole_job"] == (8000000 if scaling == "weak" else 4000000)
            assert "count vector of all" in d["config"]["parity"]
            _check_device_scopes(d, "0,1")
    else:
        r = _run(["--gpus", str(have + 1), "--reads", "1000000"])
        assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_scopes_b_and_e_over_several_devices_from_one_rank():
    """FQTK_BENCH_DEVICES=0,0: what rank 0 of an N > 1 job does for scopes B and E, on a one-GPU box (two matchers / two record
    pipelines on the same device)."""
    d = _line(_run(["--reads", "3000000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--e2e-templates", "4000000", "--e2e-threads", "8"],
                   env={"FQTK_BENCH_DEVICES": "0,0"}))
    assert d["n_gpus"] == 1 and d["scope"] == "K"
    _check_device_scopes(d, "0,0")


def test_gpus_n_without_gpus_refuses_loudly_and_prints_no_json_line():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs present")
    r = _run(["--gpus", "8", "--reads", "1000"], timeout=300)
    assert r.returncode == 2 and not r.stdout.strip()
