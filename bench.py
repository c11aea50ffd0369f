#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native fqtk demux barcode matcher.

Metric (BASELINE.json): M reads/sec demuxed (bit-exact assigns) + achieved HBM GB/s vs peak.
Workload at N=1: BASELINE.json configs[2] -- the config the north_star target is quoted on -- dual-index
400 M reads x 384 samples (8+8 bp), max-mismatches 1, min-mismatch-delta 2; it fits one GPU
(6.4 GB of observed barcodes + 1.6 GB of results in HBM).  A "step" = one pass of the hot path
(fqtk_matcher_assign_batch_device through the C ABI) over the rank's HBM-resident batch.
Multi-GPU: reads shard across ranks with no data-path collective (weak scaling: every rank owns a
full batch); the only collective is the final per-sample count all-reduce over RCCL.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--reads R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _cpu_all_cores_worker(job):
    """One host process of the all-cores row: its own matcher (memo cache on), its own slice of the stream."""
    cfg_id, mm, delta, start, seconds = job
    import dataclasses as _dc
    from fqtk_amd import synth
    from oracle import oracle as O
    cfg = _dc.replace(synth.CONFIGS[cfg_id], max_mismatches=mm, min_mismatch_delta=delta)
    w = synth.Workload(synth.CONFIGS[cfg_id])
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True, native=True)
    L, chunk, done, t = cfg.barcode_len, 500_000, 0, 0.0
    while t < seconds:
        host = np.ascontiguousarray(w.fill_host(start + done, chunk)[:, :L])
        t0 = time.perf_counter()
        lit.assign_batch(host)
        t += time.perf_counter() - t0
        done += chunk
    return done, t


def cpu_all_cores(cfg_id: int, cfg, seconds: float):
    """The same oracle on MANY host cores (one process per core, each with its own memo cache and slice):
    the row to read the GPU number against when the host is not limited to the reference's one thread."""
    import multiprocessing as mp
    procs = max(1, min(64, (os.cpu_count() or 2) // 2))
    jobs = [(cfg_id, cfg.max_mismatches, cfg.min_mismatch_delta, 10_000_000 + i * 5_000_000, seconds) for i in range(procs)]
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_cpu_all_cores_worker, jobs)
    rate = sum(d / t for d, t in res)   # every process was timed on its own matcher calls only
    return {"value": round(rate / 1e6, 2), "unit": "M reads/s", "cores": procs,
            "sample": f"{procs} processes x {seconds:.0f} s of matcher time, {sum(d for d, _ in res)} reads"}


def cpu_baseline(workload, seconds: float):
    """The oracle (literal C restatement of the reference algorithm, memo cache ON exactly as
    demux.rs:925 drives it) on ONE host core, over a bounded prefix of the same synthetic workload."""
    from oracle import oracle as O
    cfg = workload.cfg
    L = cfg.barcode_len
    # one matcher for the whole sample, memo cache cold at the start, exactly like a real demux run;
    # chunks are consumed until `seconds` of matcher time have been spent (input generation untimed)
    lit = O.RefLiteral(workload.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True, native=True)
    chunk = 1_000_000
    total_t = 0.0
    done = 0
    while total_t < seconds and done < 200_000_000:
        host = np.ascontiguousarray(workload.fill_host(done, chunk)[:, :L])
        t0 = time.perf_counter()
        lit.assign_batch(host)
        total_t += time.perf_counter() - t0
        done += chunk
    hits, misses = lit.cache_stats
    return {
        "value": round(done / total_t / 1e6, 4),
        "unit": "M reads/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {done} reads of the same synthetic workload, oracle/ref_literal.c (gcc -O3 "
                  f"-march=native), memo cache on (hit rate {hits / max(hits + misses, 1):.3f}), "
                  f"{total_t:.1f} s of CPU work; host has {os.cpu_count()} logical cores",
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config id (1-5), default 3")
    ap.add_argument("--reads", type=int, default=0, help="reads per rank per step (default: the config's N)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget (0 = skip)")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the oracle on many host cores (one process per core); adds ~10 s")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--max-mismatches", type=int, default=-1, help="override the config's value (exploration only)")
    ap.add_argument("--min-mismatch-delta", type=int, default=-1, help="override the config's value (exploration only)")
    ap.add_argument("--no-cache", action="store_true",
                    help="use_cache=false: exhaustive per-sample scan for every read (no memo table)")
    ap.add_argument("--memo-table", action="store_true",
                    help="pin the HBM/L2 table form of the memo (default: LDS-resident form when it can be built)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from fqtk_amd import BarcodeMatcher, synth
    from fqtk_amd.sharding import allreduce_counts

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the matcher has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # FQTK_BENCH_FORCE_DIST=1 exercises the RCCL code path (init, barrier, all-reduce) even with one rank
    use_dist = world > 1 or bool(os.environ.get("FQTK_BENCH_FORCE_DIST"))
    if use_dist:    # one process per GPU; backend "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)

    cfg = synth.CONFIGS[args.config]
    if args.max_mismatches >= 0 or args.min_mismatch_delta >= 0:
        import dataclasses
        cfg = dataclasses.replace(cfg,
                                  max_mismatches=args.max_mismatches if args.max_mismatches >= 0 else cfg.max_mismatches,
                                  min_mismatch_delta=args.min_mismatch_delta if args.min_mismatch_delta >= 0 else cfg.min_mismatch_delta,
                                  name=cfg.name + " [overridden mismatch parameters]")
    n = args.reads or cfg.n_reads
    workload = synth.Workload(synth.CONFIGS[args.config])
    workload.cfg = cfg
    stream = torch.cuda.current_stream().cuda_stream

    # ---- inputs resident in HBM before the timed region ------------------------------------------
    d_obs = torch.empty((n, cfg.stride), dtype=torch.uint8, device=dev)
    gen_chunk = 50_000_000
    base = rank * n                                # every rank owns a distinct shard of the stream
    for lo in range(0, n, gen_chunk):
        cur = min(gen_chunk, n - lo)
        workload.fill_device(base + lo, cur, d_obs.data_ptr() + lo * cfg.stride, stream)
    d_out = torch.empty(n, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(cfg.n_samples + 1, dtype=torch.int64, device=dev)

    matcher = BarcodeMatcher(workload.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta,
                             use_cache=not args.no_cache, device=local_rank)
    if args.memo_table:
        matcher.memo_kind = BarcodeMatcher.MEMO_TABLE
    memo_on = (not args.no_cache) and matcher.memo_entries > 0
    kernel_name = {BarcodeMatcher.MEMO_NONE: "fqtk::match_kernel", BarcodeMatcher.MEMO_TABLE: "fqtk::memo_kernel",
                   BarcodeMatcher.MEMO_LDS: "fqtk::lds_memo_kernel"}[matcher.memo_kind]

    def step():
        matcher.assign_batch_device(d_obs.data_ptr(), cfg.stride, n, d_out.data_ptr(), d_counts.data_ptr(),
                                    stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    d_counts.zero_()

    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides ------------------------
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()                                   # same stream as the kernels
    total_counts = d_counts
    if use_dist:                                   # the one collective: per-sample counts over RCCL
        total_counts = allreduce_counts(d_counts.clone())
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events around the K launches, per launch
    matcher.poll_error(stream)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness gates outside the timed region -------------------------------------------------
    counts_host = d_counts.cpu().numpy()
    if not os.environ.get("FQTK_MEMO_ABLATE"):
        assert int(counts_host.sum()) == n * args.steps, "per-sample counts do not add up to reads x steps"
    if use_dist:
        assert int(total_counts.sum().item()) == n * args.steps * world
    parity = None
    if not args.no_verify and rank == 0:
        from oracle import oracle as O
        lit = O.RefLiteral(workload.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
        dt = np.dtype([("idx", "<u2"), ("best", "u1"), ("next", "u1")])
        checked = 0
        for start in (0, n // 2, max(n - 100_000, 0)):
            m = min(100_000, n - start)
            host = workload.fill_host(base + start, m)
            i, b, nx, _ = lit.assign_batch(np.ascontiguousarray(host[:, :cfg.barcode_len]))
            got = d_out[start:start + m].cpu().numpy().view(dt)
            ok = np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx)
            assert ok, f"GPU results differ from the oracle in window starting at read {start}"
            checked += m
        parity = f"bit-exact vs oracle on {checked} reads (3 windows)"

    # HBM traffic of the dominant kernel: measured separately with the PMC counters (they cannot be
    # read live) by tools/profile_bench.sh and committed under profiles/; reported only when it was
    # collected for this exact workload and kernel, else null.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            t = json.load(fh).get(f"cfg{args.config}")
        if t and t["reads_per_launch"] == n and kernel_name.startswith(t["kernel_prefix"]):
            traffic = t["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    out = None
    if rank == 0:
        reads_total = n * args.steps * world
        value = reads_total / elapsed / 1e6
        achieved = n * cfg.bytes_per_read / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "M reads/sec demuxed (bit-exact assigns)",
            "value": round(value, 2),
            "unit": "M reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": cfg.name,
                "reads_per_gpu_per_step": n,
                "samples": cfg.n_samples,
                "barcode_len": cfg.barcode_len,
                "max_mismatches": cfg.max_mismatches,
                "min_mismatch_delta": cfg.min_mismatch_delta,
                "sharding": f"reads sharded over {world} rank(s), table replicated, RCCL all-reduce of counts only",
                "parity": parity,
                "use_cache": not args.no_cache,
                "memo_entries": matcher.memo_entries,
                "memo_kind": {0: "none (scan)", 1: "table in HBM/L2 + LDS hot subset", 2: "LDS-resident"}[matcher.memo_kind],
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": traffic,
                "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                "algorithmic_bytes_per_launch": n * cfg.bytes_per_read,
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_read": cfg.bytes_per_read,
            },
        }
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(workload, args.cpu_seconds)
            if args.cpu_all_cores:
                out["cpu_baseline"]["all_cores"] = cpu_all_cores(args.config, cfg, min(5.0, args.cpu_seconds))
            out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is flushed at exit: push it out now so
        # that the JSON line is the LAST thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
