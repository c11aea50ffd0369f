"""Multi-GPU sharding of the demux hot path: one process per GPU, reads partitioned, table replicated,
and exactly ONE collective -- the end-of-run all-reduce of the per-sample count vector (S+1 counters,
the `templates` column of demux-metrics.txt; reference /root/reference/src/bin/commands/demux.rs:
970-974, 994-998).  Every template's assignment depends only on its own barcode, so the data path
needs no exchange.  Backend-agnostic: `nccl` (= RCCL over xGMI) on GPUs,
`gloo` in the CPU tests.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of n_total reads owned by `rank` (sizes differ by <= 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_owner(chunk_index: int, world: int) -> int:
    """Streaming form: chunk k of the read stream goes to GPU k mod G; writers consume chunks in
    sequence order, so per-sample output keeps input order (demux.rs:945-977 is sequential)."""
    return chunk_index % world


def allreduce_counts(counts, group=None):
    """Sums the (S+1) per-sample counters over all ranks in place and returns them.  `counts` is a
    torch int64 tensor on the backend's device.  A no-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts
