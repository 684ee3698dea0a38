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


# ---- the sharded driver (dynaboa_amd/sharded.py) with a stand-in adaptor: sequence assignment, per-frame records, ragged gather
class _FakeAdaptor:
    """excute() 'adapts' a sequence by returning metrics that encode which frames it saw, in order."""

    def __init__(self):
        from types import SimpleNamespace
        self.options, self.device = SimpleNamespace(deferred_metrics=0, batch_size=1), torch.device("cpu")

    def excute(self, frames, nframes=None):
        ids = [float(b["id"]) for b in frames]
        return dict(mpjpe=[np.array([i + 0.25]) for i in ids], pampjpe=[np.array([2 * i]) for i in ids], pve=[i + 0.5 for i in ids])


def _specs():
    from dynaboa_amd.sharded import SequenceSpec
    lens, first, out = [5, 2, 7, 1, 3], 0, []
    for k, n in enumerate(lens):
        out.append(SequenceSpec(f"seq{k}", first, n, (lambda first=first, n=n: [dict(id=first + j) for j in range(n)])))
        first += n
    return out


def _driver_worker(rank, world, port, out):
    from types import SimpleNamespace
    from dynaboa_amd.sharded import run_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = run_sharded(SimpleNamespace(batch_size=1), _specs(), _FakeAdaptor, num_shards=world, shard_rank=rank, seqs_per_gpu=1)
    out[rank] = {k: (np.asarray(v).tolist() if not isinstance(v, int) else v) for k, v in res.items()}
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_driver_world2_gathers_every_frame_once():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mp.Manager().dict()
    mp.spawn(_driver_worker, args=(2, port, out), nprocs=2, join=True)
    for r in (0, 1):
        assert out[r]["global_index"] == list(range(18))                          # every frame of every sequence, once, in stream order
        assert out[r]["mpjpe"] == [i + 0.25 for i in range(18)] and out[r]["pampjpe"] == [2.0 * i for i in range(18)]
        assert out[r]["pve"] == [i + 0.5 for i in range(18)]
    assert sorted(out[0]["owned"] + out[1]["owned"]) == [0, 1, 2, 3, 4]
    assert out[0]["frames_local"] + out[1]["frames_local"] == 18 and abs(out[0]["frames_local"] - out[1]["frames_local"]) <= 7


def test_sharded_driver_single_process_is_the_whole_stream():
    from types import SimpleNamespace
    from dynaboa_amd.sharded import run_sharded
    res = run_sharded(SimpleNamespace(batch_size=1), _specs(), _FakeAdaptor)
    assert res["global_index"].tolist() == list(range(18)) and res["owned"] == [0, 1, 2, 3, 4]
