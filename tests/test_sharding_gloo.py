"""N>1 path on CPU: world_size-2 gloo processes each own a shard of the synthetic read stream, compute
their per-sample counts (with the CPU oracle here -- there is no GPU in this container; on the GPU box
the same wiring runs over RCCL in bench.py), all-reduce them, and must reproduce the single-process
counts exactly."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from fqtk_amd.sharding import chunk_owner, shard_range  # noqa: E402


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 100, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)
    assert [chunk_owner(k, 4) for k in range(6)] == [0, 1, 2, 3, 0, 1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, out_q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from fqtk_amd import synth
    from fqtk_amd.sharding import allreduce_counts, shard_range
    from oracle import oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.CONFIGS[2]
    w = synth.Workload(cfg)
    lo, hi = shard_range(n_total, rank, world)
    obs = w.fill_host(lo, hi - lo)
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    _, _, _, c = lit.assign_batch(obs)
    t = torch.from_numpy(c.astype(np.int64))
    allreduce_counts(t)
    dist.barrier()
    out_q.put((rank, t.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_gloo_count_allreduce_matches_single_process():
    import torch.multiprocessing as mp
    from fqtk_amd import synth
    from oracle import oracle as O

    n_total, world = 60_001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = synth.CONFIGS[2]
    w = synth.Workload(cfg)
    lit = O.RefLiteral(w.barcodes, cfg.max_mismatches, cfg.min_mismatch_delta, True)
    _, _, _, ref = lit.assign_batch(w.fill_host(0, n_total))
    for r in range(world):
        assert np.array_equal(results[r].astype(np.uint64), ref)
    assert int(ref.sum()) == n_total
