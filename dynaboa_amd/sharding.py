"""Multi-GPU: the test stream shards by SEQUENCE, never inside one (every frame's update depends on
the weights / Adam state / teacher / history the previous frame left: reference
dynaboa_benchmark.py:83-103).  One process per GPU, each with a full replica that starts from the
same checkpoint and walks its own sequences; the only collective is the end-of-run gather of
per-frame errors over RCCL/xGMI.  Replicas are deliberately NOT synchronised.

Caveat (reported, not hidden): the reference's published numbers come from ONE stream over all
sequences in order, so sharding changes what each replica has adapted on; ``num_shards=1`` is the
parity mode."""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch


def assign_sequences(frame_counts: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time greedy: whole sequences to ranks, balancing frame counts.
    Returns, per rank, the sequence indices in their original (reference) order."""
    loads = [0] * world_size
    owned: List[List[int]] = [[] for _ in range(world_size)]
    for idx in sorted(range(len(frame_counts)), key=lambda i: (-frame_counts[i], i)):
        r = min(range(world_size), key=lambda k: (loads[k], k))
        owned[r].append(idx)
        loads[r] += frame_counts[idx]
    return [sorted(o) for o in owned]


def gather_frame_metrics(local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather a per-frame metric vector whose length differs per rank (RCCL has no gatherv:
    gather the counts, pad to the maximum, gather, strip).  Works on any backend (gloo in the CPU
    tests, nccl == RCCL on the GPUs).  Returns the concatenation in rank order on every rank."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.numel()], device=local.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts) if counts else 0
    padded = torch.zeros(max(m, 1), device=local.device, dtype=local.dtype)
    padded[:local.numel()] = local.flatten()
    bufs = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)])


def shard_stream(sequences: Sequence[Dict], rank: int, world_size: int) -> List[Dict]:
    """``sequences``: items with a ``'frames'`` length.  Returns this rank's sequences."""
    owned = assign_sequences([s["frames"] for s in sequences], world_size)[rank]
    return [sequences[i] for i in owned]
