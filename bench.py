#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native fqtk demux barcode matcher.

Metric (BASELINE.json): M reads/sec demuxed (bit-exact assigns) + achieved HBM GB/s vs peak.
Workload at N=1: BASELINE.json configs[2] -- the config the north_star target is quoted on -- dual-index
400 M reads x 384 samples (8+8 bp), max-mismatches 1, min-mismatch-delta 2; it fits one GPU
(6.4 GB of observed barcodes + 1.6 GB of results in HBM).  A "step" = one pass of the hot path
(fqtk_matcher_assign_batch_device through the C ABI) over the rank's HBM-resident batch.

`value` is scope K (inputs resident in HBM; the line says `"scope": "K"`).  The same JSON line carries, never
conflated with it: `scopes.B` (C ABI with pinned host buffers, PCIe inclusive), `scopes.E` (the `fqtk demux` binary,
files -> files), `create_ms` (memo build at fqtk_matcher_create) and the CPU rows C1 / cache-off / all-cores.
With N > 1 ranks (or FQTK_BENCH_DEVICES=a,b,.. on one rank: the same device may be named several times) rank 0 also runs
scope B through one matcher per device and scope E through `fqtk demux --devices a,b,..` (plain and BGZF inputs): all three
scopes per GPU count.

Multi-GPU: reads shard across ranks with no data-path collective; the only collective is the final
per-sample count all-reduce over RCCL.  `--scaling weak` (default): every rank owns a full batch;
`--scaling strong`: the config's N reads are split over the ranks (BASELINE config 3's wording).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3] [--reads R]
        --gpus N > 1 without a torchrun environment re-launches itself as N ranks:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DT = np.dtype([("idx", "<u2"), ("best", "u1"), ("next", "u1")])


# ---- CPU rows: the oracle timed on this box's host cores --------------------------
def _cpu_worker(job):
    """One host process: its own matcher, its own slice of the synthetic stream; timed on matcher calls only."""
    cfg_id, mm, delta, use_cache, start, seconds, max_reads = job
    from fqtk_amd import synth
    from oracle import oracle as O
    w = synth.Workload(synth.CONFIGS[cfg_id])
    L = w.cfg.barcode_len
    lit = O.RefLiteral(w.barcodes, mm, delta, use_cache, native=True)
    chunk = 1_000_000 if use_cache else 100_000
    done, t = 0, 0.0
    while t < seconds and done < max_reads:
        host = np.ascontiguousarray(w.fill_host(start + done, chunk)[:, :L])
        t0 = time.perf_counter()
        lit.assign_batch(host)
        t += time.perf_counter() - t0
        done += chunk
    hits, misses = lit.cache_stats
    return done, t, hits, misses


def cpu_rows(cfg_id: int, cfg, seconds: float, pool):
    """C1: cache on, 1 core, cold cache at read 0 -- exactly how demux drives the matcher (demux.rs:925,968).
    cache_off_1core: the arithmetic the GPU replaces.  all_cores: C1 on many processes."""
    done, t, hits, misses = _cpu_worker((cfg_id, cfg.max_mismatches, cfg.min_mismatch_delta, True, 0, seconds, 200_000_000))
    out = {
        "value": round(done / t / 1e6, 4), "unit": "M reads/s", "cores": 1, "kind": "port", "row": "C1",
        "sample": f"first {done} reads of the same synthetic workload, oracle/ref_literal.c (gcc -O3 -march=native), "
                  f"memo cache on (hit rate {hits / max(hits + misses, 1):.3f}), {t:.1f} s of CPU work; "
                  f"host shows {os.cpu_count()} logical CPUs, {effective_cpus()} usable (affinity / cgroup quota)",
    }
    d2, t2, _, _ = _cpu_worker((cfg_id, cfg.max_mismatches, cfg.min_mismatch_delta, False, 0, min(seconds, 5.0), 50_000_000))
    out["cache_off_1core"] = {"value": round(d2 / t2 / 1e6, 4), "unit": "M reads/s", "cores": 1, "row": "C2",
                              "sample": f"first {d2} reads, memo cache off (every read scans all samples), {t2:.1f} s"}
    if pool is not None:
        procs = pool.n_procs
        jobs = [(cfg_id, cfg.max_mismatches, cfg.min_mismatch_delta, True, 10_000_000 + i * 5_000_000, min(seconds, 5.0),
                 5_000_000) for i in range(procs)]
        res = pool.map(_cpu_worker, jobs)
        out["all_cores"] = {"value": round(sum(d / tt for d, tt, _, _ in res) / 1e6, 2), "unit": "M reads/s", "cores": procs,
                            "sample": f"{procs} processes (own matcher + memo cache + slice each) on {effective_cpus()} usable CPUs, "
                                      f"{sum(d for d, _, _, _ in res)} reads; sum of per-process rates"}
    return out


# ---- parity gate: triples on a prefix, per-sample count vector on everything ----------------
def _parity_worker(job):
    cfg_id, mm, delta, seed_base, lo, hi, triple_hi, path = job
    from fqtk_amd import synth
    from oracle import oracle as O
    w = synth.Workload(synth.CONFIGS[cfg_id])
    L = w.cfg.barcode_len
    lit = O.RefLiteral(w.barcodes, mm, delta, True, native=True)
    got = np.memmap(path, dtype=DT, mode="r") if (path and lo < triple_hi) else None
    counts = np.zeros(w.cfg.n_samples + 1, dtype=np.uint64)
    bad = 0
    for a in range(lo, hi, 2_000_000):
        b = min(hi, a + 2_000_000)
        host = np.ascontiguousarray(w.fill_host(seed_base + a, b - a)[:, :L])
        i, be, nx, c = lit.assign_batch(host)
        counts += c
        if got is not None and a < triple_hi:
            m = min(b, triple_hi) - a
            g = got[a:a + m]
            bad += int(np.count_nonzero((g["idx"] != i[:m]) | (g["best"] != be[:m]) | (g["next"] != nx[:m])))
    return bad, counts


def parity_gate(pool, cfg_id, cfg, total_reads, triple_reads, d_out, gpu_counts_per_step):
    """Oracle fanned over host processes: every (idx, best, next) of the first `triple_reads` reads of rank 0
    and the per-sample counts of all `total_reads` reads of the job (all ranks' shards are one stream)."""
    path = f"/dev/shm/fqtk_bench_parity_{os.getpid()}.bin"
    d_out[:triple_reads].cpu().numpy().tofile(path)
    try:
        procs = pool.n_procs
        bounds = np.linspace(0, total_reads, procs * 4 + 1).astype(np.int64)
        jobs = [(cfg_id, cfg.max_mismatches, cfg.min_mismatch_delta, 0, int(bounds[i]), int(bounds[i + 1]), triple_reads, path)
                for i in range(len(bounds) - 1) if bounds[i] < bounds[i + 1]]
        t0 = time.perf_counter()
        res = pool.map(_parity_worker, jobs, chunksize=1)
        wall = time.perf_counter() - t0
    finally:
        os.unlink(path)
    bad = sum(r[0] for r in res)
    oracle_counts = sum((r[1] for r in res), np.zeros(cfg.n_samples + 1, dtype=np.uint64))
    ok_counts = bool(np.array_equal(oracle_counts, gpu_counts_per_step))
    assert bad == 0, f"{bad} reads of the first {triple_reads} differ from the oracle"
    assert ok_counts, "per-sample count vector differs from the oracle's"
    return (f"bit-exact vs oracle: all (idx,best,next) of the first {triple_reads} reads + the per-sample count vector "
            f"of all {total_reads} reads ({procs} host processes, {wall:.1f} s)")


def pmc_traffic(cfg_id: int, n: int, kernel_name: str):
    """HBM bytes per launch of the dominant kernel from the PMC counters: they cannot be read live, so
    tools/profile_bench.sh collects them (separate --pmc passes, gfx950 correction) and records the digest of
    the kernel sources they were measured on; a kernel edited since then reports null, not a stale number."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            t = json.load(fh).get(f"cfg{cfg_id}")
        if t and t["reads_per_launch"] == n and kernel_name.startswith(t["kernel_prefix"]) \
                and t.get("kernel_sources_sha1") == kernel_sources_digest():
            return t["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def effective_cpus() -> int:
    """Logical CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (cpu.max / cfs_quota) -- the GPU boxes of this pool show 256 logical CPUs under a 16-CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


MATCHER_KERNEL_SOURCES = ("match_kernels.hip.h", "memo_kernels.hip.h", "lds_memo_kernels.hip.h", "memo_hash.hpp",
                          "lds_memo_plan.hpp", "direct_memo_plan.hpp", "fqtk_match.hip")


def kernel_sources_digest() -> str:
    """Digest of the files that define the matcher's kernels, their tables and their launch shapes."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "fqtk_amd", "csrc")
    for f in MATCHER_KERNEL_SOURCES:
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def self_launch(n_gpus: int) -> int:
    """--gpus N without a torchrun environment: start N ranks (one process per GPU) and relay the result."""
    import torch
    have = torch.cuda.device_count()
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} but only {have} GPU(s) are visible; refusing to report a smaller job "
              f"under that label", file=sys.stderr)
        return 2
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, env=env).returncode


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of 70 ms on cfg 3.  The device's clocks take tens of milliseconds to settle after the idle stretch of the
    # matcher's creation: 10 steps after 2 warm-up steps (the defaults until the end of round 6) read 1-3 % under the rate of 200 or 1000
    # steps on the same box, 50 after 10 read the same (profiles/r06_bench_window.txt)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config id (1-5), default 3")
    ap.add_argument("--reads", type=int, default=0, help="reads per rank per step (default: the config's N)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank owns --reads reads; strong: they are split over the ranks")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline budget for row C1 (0 = skip all CPU rows)")
    ap.add_argument("--parity", choices=("full", "windows", "none"), default=None,
                    help="full: first 50 M triples + count vector of all reads (default at 1 rank); windows: 3 x 100 k")
    ap.add_argument("--no-verify", action="store_true", help="same as --parity none")
    ap.add_argument("--no-scopes", action="store_true", help="skip scopes B and E")
    ap.add_argument("--e2e-templates", type=int, default=64_000_000,
                    help="templates of the scope E run (multiples of 1 M above 1 M: the first 1 M templates repeated)")
    ap.add_argument("--e2e-threads", type=int, default=0, help="--threads of the scope E run (default: the usable CPUs, at most 32)")
    ap.add_argument("--e2e-gz", action="store_true", help="gzip the scope E inputs (single-stream gunzip per file)")
    ap.add_argument("--lens", action="store_true", help="pass an obs_len array (all == L): the variable-length '+B' path")
    ap.add_argument("--max-mismatches", type=int, default=-1, help="override the config's value (exploration only)")
    ap.add_argument("--min-mismatch-delta", type=int, default=-1, help="override the config's value (exploration only)")
    ap.add_argument("--no-cache", action="store_true",
                    help="use_cache=false: exhaustive per-sample scan for every read (no memo table)")
    ap.add_argument("--memo-table", action="store_true",
                    help="pin the HBM/L2 table form of the memo (default: LDS-resident form when it can be built)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)

    import torch
    import torch.distributed as dist

    from fqtk_amd import BarcodeMatcher, synth
    from fqtk_amd.sharding import allreduce_counts, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the matcher has no CPU fallback", file=sys.stderr)
        return 2
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}", file=sys.stderr)
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
        assert dist.get_world_size() == args.gpus, "process group size != --gpus"

    cfg = synth.CONFIGS[args.config]
    if args.max_mismatches >= 0 or args.min_mismatch_delta >= 0:
        import dataclasses
        cfg = dataclasses.replace(cfg,
                                  max_mismatches=args.max_mismatches if args.max_mismatches >= 0 else cfg.max_mismatches,
                                  min_mismatch_delta=args.min_mismatch_delta if args.min_mismatch_delta >= 0 else cfg.min_mismatch_delta,
                                  name=cfg.name + " [overridden mismatch parameters]")
    job_reads = (args.reads or cfg.n_reads) * (world if args.scaling == "weak" else 1)   # one step of the whole job
    base, hi = shard_range(job_reads, rank, world)     # this rank's contiguous shard of the read stream
    n = hi - base
    workload = synth.Workload(synth.CONFIGS[args.config])
    workload.cfg = cfg
    stream = torch.cuda.current_stream().cuda_stream

    # ---- inputs resident in HBM before the timed region ------------------------------------------
    d_obs = torch.empty((n, cfg.stride), dtype=torch.uint8, device=dev)
    gen_chunk = 50_000_000
    for lo in range(0, n, gen_chunk):
        cur = min(gen_chunk, n - lo)
        workload.fill_device(base + lo, cur, d_obs.data_ptr() + lo * cfg.stride, stream)
    d_out = torch.empty(n, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(cfg.n_samples + 1, dtype=torch.int64, device=dev)
    d_lens = torch.full((n,), cfg.barcode_len, dtype=torch.int32, device=dev) if args.lens else None

    # ---- fqtk_matcher_create: table upload + complete-memo build (enumerate, scan on the device, place) ----
    BarcodeMatcher(workload.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, device=local_rank).close()  # HIP context, code objects
    t0 = time.perf_counter()
    matcher = BarcodeMatcher(workload.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta,
                             use_cache=not args.no_cache, device=local_rank)
    create_ms = (time.perf_counter() - t0) * 1e3
    if args.memo_table:
        matcher.memo_kind = BarcodeMatcher.MEMO_TABLE
    kernel_name = {BarcodeMatcher.MEMO_NONE: "fqtk::match_kernel", BarcodeMatcher.MEMO_TABLE: "fqtk::memo_kernel",
                   BarcodeMatcher.MEMO_LDS: "fqtk::lds_memo_kernel"}[matcher.memo_kind]

    def step():
        matcher.assign_batch_device(d_obs.data_ptr(), cfg.stride, n, d_out.data_ptr(), d_counts.data_ptr(),
                                    d_lens=d_lens.data_ptr() if d_lens is not None else 0, stream=stream)

    # (untimed) launches until the device's clocks have settled after the idle stretch of the matcher's creation: 60 ms of them, less the
    # warm-up steps asked for (profiles/r06_bench_window.txt: 14 ms of cfg 3, or 50 steps of cfg 2's 0.2 ms, read 1-15 % under the sustained rate)
    step()
    torch.cuda.synchronize()                  # (the first launch: lazy initialisation)
    t_one = time.perf_counter()
    step()
    torch.cuda.synchronize()
    t_one = max(time.perf_counter() - t_one, 1e-5)
    settle_steps = min(2000, max(0, int(0.06 / t_one) - args.warmup))
    for _ in range(settle_steps):
        step()
    torch.cuda.synchronize()
    settle_steps += 2
    for _ in range(args.warmup):
        step()
        torch.cuda.synchronize()   # (untimed) a matcher adapts between batches -- the worklist for reads with IUPAC /
                                   # junk bytes is attached once one has been seen -- so warm-up steps run one at a time
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

    # ---- several ranks: the same K steps once more over 1 / world of the batch -- what every rank does when the
    #      config's N reads are SPLIT over the ranks (BASELINE config 3's wording: 400 M reads over 1 -> 8 GPUs); the
    #      line's `value` stays the weak-scaling one (every rank a full batch) unless --scaling strong was asked for
    strong = None
    if use_dist and args.scaling == "weak":
        ns = max(1, (args.reads or cfg.n_reads) // world)
        d_counts_s = torch.zeros_like(d_counts)
        dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            matcher.assign_batch_device(d_obs.data_ptr(), cfg.stride, ns, d_out.data_ptr(), d_counts_s.data_ptr(),
                                        d_lens=d_lens.data_ptr() if d_lens is not None else 0, stream=stream)
        allreduce_counts(d_counts_s.clone())
        torch.cuda.synchronize()
        dist.barrier()
        es = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=dev)
        dist.all_reduce(es, op=dist.ReduceOp.MAX)
        strong = {"value": round(ns * world * args.steps / float(es.item()) / 1e6, 2), "unit": "M reads/s",
                  "ms_per_step": round(float(es.item()) / args.steps * 1e3, 4), "reads_per_gpu_per_step": ns,
                  "reads_per_step_whole_job": ns * world,
                  "what": "strong scaling: the config's reads split over the ranks (each rank: the first 1/world of its resident "
                          "batch), barrier + synchronize on both sides, max over ranks, RCCL all-reduce of the counts included"}
        matcher.poll_error(stream)

    per_rank_kernel_ms = [round(kernel_ms, 4)]
    per_rank_reads = [n]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.zeros(world, dtype=torch.float64, device=dev)
        km[rank] = kernel_ms
        dist.all_reduce(km)
        per_rank_kernel_ms = [round(float(x), 4) for x in km.tolist()]
        nr = torch.zeros(world, dtype=torch.int64, device=dev)
        nr[rank] = n
        dist.all_reduce(nr)
        per_rank_reads = [int(x) for x in nr.tolist()]   # an N > 1 record checks itself: sum == reads_per_step_whole_job
        assert sum(per_rank_reads) == job_reads

    # ---- correctness gates outside the timed region -------------------------------------------------
    counts_host = d_counts.cpu().numpy()
    if not os.environ.get("FQTK_MEMO_ABLATE"):
        assert int(counts_host.sum()) == n * args.steps, "per-sample counts do not add up to reads x steps"
    job_counts = total_counts.cpu().numpy().astype(np.uint64)
    if use_dist:
        assert int(job_counts.sum()) == job_reads * args.steps, "all-reduced counts do not add up to the job's reads"
    assert np.all(job_counts % args.steps == 0), "the K passes over the same batch disagree with each other"
    mode = "none" if args.no_verify else (args.parity or ("full" if world == 1 else "windows"))
    host_cores = effective_cpus()
    pool = None
    if rank == 0 and (mode == "full" or (world == 1 and args.cpu_seconds > 0)):
        import multiprocessing as mp
        from oracle import oracle as O
        O.build(native=True)                       # once, before the workers all ask for it
        n_procs = max(1, min(128, host_cores))
        pool = mp.get_context("spawn").Pool(n_procs)
        pool.n_procs = n_procs
    parity = None
    if rank == 0 and mode == "full":
        # the oracle must finish in about a minute: on a small host the count vector covers a prefix only
        budget = job_reads if host_cores >= 12 else min(job_reads, 50_000_000)
        if budget == job_reads:
            parity = parity_gate(pool, args.config, cfg, job_reads, min(n, 50_000_000), d_out, job_counts // args.steps)
        else:
            mode = "windows"
    if rank == 0 and mode == "windows":
        from oracle import oracle as O
        lit = O.RefLiteral(workload.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
        checked = 0
        for start in (0, n // 2, max(n - 100_000, 0)):
            m = min(100_000, n - start)
            host = workload.fill_host(base + start, m)
            i, b, nx, _ = lit.assign_batch(np.ascontiguousarray(host[:, :cfg.barcode_len]))
            got = d_out[start:start + m].cpu().numpy().view(DT)
            ok = np.array_equal(got["idx"], i) and np.array_equal(got["best"], b) and np.array_equal(got["next"], nx)
            assert ok, f"GPU results differ from the oracle in window starting at read {start}"
            checked += m
        parity = f"bit-exact vs oracle on {checked} reads (3 windows of rank 0's shard)"

    # the ranks are done with each other: the process group goes NOW, on every rank at once (rank 0 goes on alone with the scopes and
    # the CPU rows, the others leave)
    rccl_ranks = dist.get_world_size() if use_dist else 1
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    out = None
    if rank == 0:
        value = job_reads * args.steps / elapsed / 1e6
        bytes_per_read = cfg.bytes_per_read + (4 if args.lens else 0)   # an obs_len batch reads one more dword per read
        achieved = n * bytes_per_read / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "M reads/sec demuxed (bit-exact assigns)",
            "value": round(value, 2),
            "unit": "M reads/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,   # the process group the counts were all-reduced over
            "reads_per_step_per_rank": per_rank_reads,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            **({"strong_scaling": strong} if strong is not None else {}),
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "scope": "K: observed barcodes resident in HBM, results left in HBM (SURVEY.md 8d); B and E are in `scopes`",
            "config": {
                "workload": cfg.name,
                "reads_per_gpu_per_step": n,
                "reads_per_step_whole_job": job_reads,
                "samples": cfg.n_samples,
                "barcode_len": cfg.barcode_len,
                "max_mismatches": cfg.max_mismatches,
                "min_mismatch_delta": cfg.min_mismatch_delta,
                "sharding": f"reads sharded over {world} rank(s), table replicated, RCCL all-reduce of counts only",
                "parity": parity,
                "use_cache": not args.no_cache,
                "obs_len": bool(args.lens),
                "memo_entries": matcher.memo_entries,
                "memo_kind": {0: "none (scan)", 1: "table in HBM/L2 + LDS hot subset", 2: "LDS-resident"}[matcher.memo_kind] +
                             (f", direct-indexed ({matcher.memo_direct_bytes}-byte entries) for reads without a no-call"
                              if matcher.memo_kind == 1 and matcher.memo_direct_bytes else ""),
            },
            "create_ms": round(create_ms, 2),
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": pmc_traffic(args.config, n, kernel_name),
                "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                "algorithmic_bytes_per_launch": n * bytes_per_read,
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_per_rank": per_rank_kernel_ms,
                "algorithmic_bytes_per_read": bytes_per_read,
            },
        }
        dev_list = list(range(world)) if world > 1 else None
        if world == 1 and os.environ.get("FQTK_BENCH_DEVICES"):   # one rank, several record pipelines / matchers (a one-GPU box: "0,0")
            dev_list = [int(x) for x in os.environ["FQTK_BENCH_DEVICES"].split(",")]
        if dev_list is not None and len(dev_list) > 1 and not args.no_scopes:
            # ---- scopes B and E over all the job's devices, from this one process (the other ranks are done with their GPUs) ----
            import scope_bench
            del d_obs, d_out
            torch.cuda.empty_cache()
            out["scopes"] = device_scopes(scope_bench, args, dev_list, workload, value, pool, host_cores)
        elif world == 1 and not args.no_scopes:
            import scope_bench
            scopes = {"K": {"M_reads_per_s": round(value, 1), "what": "this line's `value`"}}
            del d_obs, d_out
            torch.cuda.empty_cache()
            scopes["B"] = scope_bench.scope_b(args.config, matcher=matcher, workload=workload,
                                              n_chunk=min(8_000_000, max(job_reads // 4, 1)))
            scopes["B_packed"] = scope_bench.scope_b_packed(args.config, matcher=matcher, workload=workload,
                                                            n_chunk=min(8_000_000, max(job_reads // 4, 1)))
            import bgzf_bench   # (tools/)
            scopes["bgzf_kernel"] = bgzf_bench.measure(blocks=2048, reps=3, where_list=("hbm",))
            scopes["bgzf_kernel"]["what"] = ("fqtk::bgzf::deflate_kernel alone: 2048 BGZF blocks of Illumina-style text in HBM -> DEFLATE payloads + CRC-32 in HBM "
                                             "(the compressor of scope E's output path, BgzfCompressor demux.rs:755-798); ratio = output / input")
            import inflate_bench   # (tools/)
            # (24 576 members: a wavefront each, four rounds of the 6 144 wavefronts the chip holds -- 4 096, rounds 4-5's size, is two thirds of ONE round)
            scopes["inflate_kernel"] = inflate_bench.measure(members=24576, reps=3)
            scopes["inflate_kernel"]["what"] = ("fqtk::inflate::inflate_kernel + member_check_kernel alone: 24576 BGZF members (zlib -6, Illumina-style text) in HBM -> text, "
                                                "CRC-32 / ISIZE check and newline counts in HBM (the decoder of scope E_bgzf's input path, demux.rs:844-849)")
            tmp = scope_bench.scratch_dir(args.e2e_templates * 900)
            try:
                expect = None
                n_e = args.e2e_templates
                rep = n_e > 1_000_000 and n_e % 1_000_000 == 0       # a 1 M-template block written n_e / 1 M times
                uniq = 1_000_000 if rep else n_e
                if pool is not None and args.config == 3:   # metrics file vs the oracle's count vector
                    res = pool.map(_parity_worker, [(3, 1, 2, 0, lo, min(lo + 250_000, uniq), 0, None)
                                                    for lo in range(0, uniq, 250_000)])
                    expect = sum((r[1] for r in res), np.zeros(385, dtype=np.uint64)) * np.uint64(n_e // uniq)
                e_threads = args.e2e_threads or max(5, min(32, host_cores))
                # E: the default run -- text to the device, whole BGZF members back (include/fqtk_demux.h)
                scopes["E"] = scope_bench.scope_e(n_e, e_threads, args.e2e_gz, tmp, expect, repeat_first_block=rep)
                scopes["E"]["host_cpus_usable"] = host_cores
                if rep and not args.e2e_gz and n_e >= 4_000_000 and n_e % 4_000_000 == 0:
                    paths = [os.path.join(tmp, x) for x in ("R1.fastq", "I1.fastq", "I2.fastq", "R2.fastq")]
                    meta = os.path.join(tmp, "meta.tsv")
                    # a quarter of the same inputs for the reference's division of labour (--host-output: host threads parse, format and
                    # libdeflate-compress; 3 M templates/s)
                    sub = scope_bench.prefix_inputs(tmp, paths, 1, 4)
                    exp4 = None if expect is None else expect // np.uint64(4)
                    scopes["E_host"] = scope_bench.scope_e(n_e // 4, e_threads, False, tmp, exp4, extra_args=("--host-output",), inputs=(sub, meta))
                    scopes["E_host"]["host_cpus_usable"] = host_cores
                    shutil.rmtree(os.path.dirname(sub[0]), ignore_errors=True)
                    # compressed inputs at the FULL size of row E (16 M-template runs are a second long and mostly start-up):
                    # the same text as one gzip member per file and as BGZF; the plain files go first (scratch is RAM)
                    gz = scope_bench.gzip_single_stream(paths)
                    bgz = scope_bench.bgzf_repeated(paths)
                    for q in paths:
                        os.unlink(q)
                    # single-stream gzip inputs: decoded on the device in chunks (the default from 64 MB of .gz), and by the host's decoders
                    scopes["E_gz"] = scope_bench.scope_e(n_e, e_threads, True, tmp, expect, inputs=(gz, meta))
                    scopes["E_gz"]["host_cpus_usable"] = host_cores
                    scopes["E_gz"]["gz_inputs"] = ("one gzip member per file (level 1): block starts found on the device (a lane per bit position), chunks of 64 KiB decoded by a wavefront each without "
                                                   "their windows, windows and CRC-32 resolved on the device")
                    scopes["E_gz_host"] = scope_bench.scope_e(n_e, e_threads, True, tmp, expect, extra_args=("--host-inflate",), inputs=(gz, meta), out_name="out_gzhost")
                    scopes["E_gz_host"]["host_cpus_usable"] = host_cores
                    scopes["E_gz_host"]["gz_inputs"] = "the same files decoded by several host threads per file (host/parallel_gunzip.hpp)"
                    # BGZF inputs (bgzip / htslib / fqtk's own outputs): the members cross PCIe compressed and are inflated on the
                    # device, one wavefront per member (include/fqtk_inflate.h, fqtk_demuxer_feed); the host never sees the text
                    scopes["E_bgzf"] = scope_bench.scope_e(n_e, e_threads, "bgzf", tmp, expect, inputs=(bgz, meta))
                    scopes["E_bgzf"]["host_cpus_usable"] = host_cores
                    scopes["E_bgzf"]["gz_inputs"] = "BGZF (65 280-byte members, level 1), inflated on the device"
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
            out["scopes"] = scopes
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_rows(args.config, cfg, args.cpu_seconds, pool)
            if "scopes" in out:   # north_star's >= 10x target is a scope-B statement against row C1
                out["cpu_baseline"]["scope_B_over_C1"] = round(out["scopes"]["B"]["M_reads_per_s"] / out["cpu_baseline"]["value"], 1)
    if pool is not None:
        pool.close()
        pool.join()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is flushed at exit: push it out now so
        # that the JSON line is the LAST thing on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(one_line(out)), flush=True)
    return 0


def device_scopes(scope_bench, args, dev_list, workload, value_k, pool, host_cores):
    """Scopes B and E with the job's devices: B = one matcher, one host thread and two page-locked slots per device;
    E = ONE `fqtk demux --devices a,b,..` process -- chunk k runs on device k mod G; plain inputs are read by the host's reader
    threads, BGZF inputs are inflated on their home devices and go to the chunk's device over xGMI -- files to files, its metrics
    file checked against the oracle's counts."""
    devs = ",".join(str(d) for d in dev_list)
    scopes = {"K": {"M_reads_per_s": round(value_k, 1), "what": "this line's `value`"}, "devices": devs}
    scopes["B"] = scope_bench.scope_b_devices(args.config, dev_list, n_chunk=8_000_000, workload=workload)
    n_e = args.e2e_templates
    tmp = scope_bench.scratch_dir(n_e * 900)
    try:
        rep = n_e > 1_000_000 and n_e % 1_000_000 == 0
        uniq = 1_000_000 if rep else n_e
        expect = None
        if pool is None and args.config == 3:
            import multiprocessing as mp
            from oracle import oracle as O
            O.build(native=True)
            pool = mp.get_context("spawn").Pool(max(1, min(32, host_cores)))
            own_pool = True
        else:
            own_pool = False
        if pool is not None and args.config == 3:
            res = pool.map(_parity_worker, [(3, 1, 2, 0, lo, min(lo + 250_000, uniq), 0, None) for lo in range(0, uniq, 250_000)])
            expect = sum((r[1] for r in res), np.zeros(385, dtype=np.uint64)) * np.uint64(n_e // uniq)
        if own_pool:
            pool.close()
            pool.join()
        e_threads = args.e2e_threads or max(5, min(32, host_cores))
        scopes["E"] = scope_bench.scope_e(n_e, e_threads, False, tmp, expect, extra_args=("--devices", devs), repeat_first_block=rep)
        scopes["E"]["host_cpus_usable"] = host_cores
        if rep:
            paths = [os.path.join(tmp, x) for x in ("R1.fastq", "I1.fastq", "I2.fastq", "R2.fastq")]
            meta = os.path.join(tmp, "meta.tsv")
            bgz = scope_bench.bgzf_repeated(paths)
            for q in paths:
                os.unlink(q)
            scopes["E_bgzf"] = scope_bench.scope_e(n_e, e_threads, "bgzf", tmp, expect, extra_args=("--devices", devs), inputs=(bgz, meta))
            scopes["E_bgzf"]["host_cpus_usable"] = host_cores
            scopes["E_bgzf"]["gz_inputs"] = "BGZF (65 280-byte members, level 1): every input inflated on its home device, chunk k's text to device k mod G device to device"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return scopes


def one_line(out: dict) -> dict:
    """The ONE JSON line, short enough for whoever keeps only the end of this program's output (the driver keeps 2000 characters):
    every field of the contract, the scopes as numbers.  The whole record -- what every scope ran, its stages and timelines, the
    CPU rows' samples -- goes to gpurun_out/bench_detail.json (profiles/ keeps copies of the runs quoted in DESIGN.md)."""
    detail = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(detail), exist_ok=True)
        with open(detail, "w") as fh:
            json.dump(out, fh, indent=1)
    except OSError:
        detail = None
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "settle_steps", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data") if k in out}
    line["scope"] = "K"   # what `value` is: barcodes resident in HBM, results left there (B and E: `scopes`)
    for k in ("rccl_ranks", "create_ms"):
        if k in out:
            line[k] = out[k]
    if "strong_scaling" in out:
        line["strong_scaling"] = out["strong_scaling"]
    c = out["config"]
    line["config"] = {"workload": c["workload"], "reads_per_step_whole_job": c["reads_per_step_whole_job"], "samples": c["samples"], "barcode_len": c["barcode_len"],
                      "max_mismatches": c["max_mismatches"], "min_mismatch_delta": c["min_mismatch_delta"], "memo": c["memo_kind"].split(",")[0],
                      "parity": (c["parity"] or "")[:160]}
    r = out["roofline"]
    line["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_bytes_per_launch")}
    if out.get("n_gpus", 1) > 1:
        line["roofline"]["kernel_ms_per_rank"] = r.get("kernel_ms_per_rank")
    if "cpu_baseline" in out:
        b = out["cpu_baseline"]
        line["cpu_baseline"] = {"value": b["value"], "unit": b["unit"], "cores": b["cores"], "kind": b["kind"],
                                "sample": "first %s reads, oracle/ref_literal.c, cache on" % b["sample"].split(" ")[1],
                                "cache_off": b.get("cache_off_1core", {}).get("value"), "all_cores": [b.get("all_cores", {}).get("value"), b.get("all_cores", {}).get("cores")]}
        if "scope_B_over_C1" in b:
            line["cpu_baseline"]["scope_B_over_C1"] = b["scope_B_over_C1"]
    if "scopes" in out:
        sc, short = out["scopes"], {}
        for k in ("B", "B_packed"):
            if k in sc:
                short[k] = sc[k]["M_reads_per_s"]
        for k in ("E", "E_host", "E_gz", "E_gz_host", "E_bgzf"):   # [M templates/s wall clock of the process, steady, M templates]
            if k in sc:
                short[k] = [sc[k]["M_templates_per_s"], sc[k]["M_templates_per_s_steady"], round(sc[k]["templates"] / 1e6)]
        if "bgzf_kernel" in sc:      # GB/s of text in (deflate_kernel alone, blocks in HBM)
            short["bgzf_kernel_GBps"] = sc["bgzf_kernel"].get("hbm", {}).get("GB_per_s_in")
        if "inflate_kernel" in sc:   # GB/s of text out (inflate_kernel + check alone)
            short["inflate_kernel_GBps"] = sc["inflate_kernel"].get("text_GBps")
        if "devices" in sc:          # B and E* ran over these devices (N > 1 ranks, or FQTK_BENCH_DEVICES)
            short["devices"] = sc["devices"]
        short["is"] = "B*: M reads/s host->host; E*: fqtk demux files->files [M templates/s wall, steady, M templates], counts = oracle's"
        line["scopes"] = short
    if detail:
        line["detail"] = os.path.relpath(detail, ROOT)
    return line


if __name__ == "__main__":
    rc = main()
    # The line is printed and every file is written.  Unless a profiler needs the process's orderly end (rocprofv3 writes its
    # traces there) or FQTK_CLEAN_EXIT=1 asks for it, leave without the interpreter's teardown of torch and the HIP runtime: a sibling
    # tool's ended in SIGSEGV once in some forty runs on the pool's boxes (round 6), after its result had been printed.
    profiled = any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if profiled or os.environ.get("FQTK_CLEAN_EXIT", "0") not in ("", "0"):
        sys.exit(rc)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(rc or 0)
