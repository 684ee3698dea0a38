"""N>1 path on CPU: sequence sharding and the end-of-stream ragged gather over gloo, world_size 2."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dynaboa_amd.sharding import assign_sequences, gather_frame_metrics, shard_stream


def test_assign_sequences_balanced_and_complete():
    counts = [1824, 919, 611, 1403, 765, 540, 880, 1000, 372, 1387, 2000, 50, 432, 1100, 903, 777]
    for world in (1, 2, 4, 8):
        owned = assign_sequences(counts, world)
        flat = sorted(i for o in owned for i in o)
        assert flat == list(range(len(counts)))                       # every sequence exactly once
        loads = [sum(counts[i] for i in o) for o in owned]
        assert max(loads) - min(loads) <= max(counts)                 # LPT bound
        assert all(o == sorted(o) for o in owned)                     # reference order kept inside a rank
    seqs = [dict(name=f"s{i}", frames=c) for i, c in enumerate(counts)]
    assert [s["name"] for s in shard_stream(seqs, 0, 1)] == [s["name"] for s in seqs]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 5 if rank == 0 else 3                                         # ragged: ranks own different frame counts
    local = torch.arange(n, dtype=torch.float32) + 100 * rank
    g = gather_frame_metrics(local)
    out[rank] = g.numpy().tolist()
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_gather_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    want = [0, 1, 2, 3, 4, 100, 101, 102]
    assert out[0] == want and out[1] == want


def test_gather_is_identity_without_process_group():
    x = torch.arange(4.0)
    assert torch.equal(gather_frame_metrics(x), x)
