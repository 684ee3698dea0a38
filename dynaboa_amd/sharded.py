"""The sharded driver: the 3DPW test stream split by SEQUENCE over processes (one per GPU) and, inside a process, over sequence
replicas stepped in lockstep (dynaboa_amd.native_step.ReplicaGroup).

Reference: dynaboa_benchmark.py:83-123 walks ONE stream over all ``3dpw_<seq>_<pid>.npz`` files in order (boa_dataset/pw3d.py:19-35);
every frame's update depends on what the previous frame left, so the only parallel axis is the sequence.  Here
  * ``--num_shards N --shard_rank r`` (default 1 / 0 = the reference's single stream, the parity mode) assigns whole sequences to
    ranks with dynaboa_amd.sharding.assign_sequences over the files in the reference's order;
  * ``--seqs_per_gpu S`` adapts S of a rank's sequences at once, each with its own weights / Adam state / teacher / history /
    records (starting from the same checkpoint), in waves of S; sequences have different lengths - one whose stream has ended
    leaves the launch set (dyb_stepper_set_active), the others are not held back;
  * at the end every rank holds (global frame index, mpjpe, pampjpe, pve) of all frames: ONE ragged all_gather (RCCL over xGMI on
    the GPUs, gloo in the CPU tests), no collective on the data path.
With N > 1 or S > 1 every sequence starts from the checkpoint - what each has adapted on differs from the reference's single
stream, stated in DESIGN.md 6."""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from .sharding import assign_sequences, gather_frame_metrics


class SequenceSpec:
    """One sequence of the stream: `frames()` yields its batches in order, `first` is the global index of its first frame."""

    def __init__(self, name: str, first: int, nframes: int, frames: Callable[[], object]):
        self.name, self.first, self.nframes, self.frames = name, int(first), int(nframes), frames


def pw3d_sequences(options, device) -> List[SequenceSpec]:
    """The reference's stream as sequences: one per 3dpw_<seq>_<pid>.npz, in the reference's file order."""
    from . import datasets as D
    ds = D.PW3D(options, img_dir=getattr(options, "pw3d_root", None) or D.PW3D_ROOT, device=device)
    out = []
    for s in ds.sequences:
        idx = list(range(s["first"], s["first"] + s["frames"]))
        out.append(SequenceSpec(s["file"], s["first"], s["frames"],
                                (lambda idx=idx: D.FrameLoader(ds, batch_size=options.batch_size, workers=4, indices=idx))))
    return out


def run_sharded(options, sequences: Sequence[SequenceSpec], make_adaptor: Callable[[], object], num_shards: int = 1, shard_rank: int = 0,
                seqs_per_gpu: int = 1, group=None) -> Dict[str, np.ndarray]:
    """Adapt this rank's sequences (waves of `seqs_per_gpu` in lockstep) and gather every rank's per-frame errors.
    make_adaptor() -> a fresh dynaboa_amd.benchmark.Adaptor at the base checkpoint.
    -> dict(global_index, mpjpe, pampjpe, pve): arrays over ALL frames of ALL ranks, sorted by global index, plus 'owned'
    (this rank's sequence indices) and 'frames_local'."""
    from . import native_step as NS
    owned = assign_sequences([s.nframes for s in sequences], num_shards)[shard_rank]
    B = int(getattr(options, "batch_size", 1))
    rows: List[np.ndarray] = []                      # (global frame index, mpjpe, pampjpe, pve) per frame
    # longest first inside a rank: a wave's sequences then have similar lengths (fewer idle replica slots at its tail)
    order = sorted(owned, key=lambda i: (-sequences[i].nframes, i))
    S = max(1, int(seqs_per_gpu))
    if S > 1 and B > 1 and any(sequences[i].nframes % B for i in owned):
        # a replica group stages B samples per sequence and step: a last, partial batch would be read past its end
        raise ValueError("seqs_per_gpu > 1 needs sequence lengths that are multiples of batch_size (or batch_size 1)")
    for w0 in range(0, len(order), S):
        wave = [sequences[i] for i in order[w0:w0 + S]]
        steps = max(-(-s.nframes // B) for s in wave)
        ads = [make_adaptor() for _ in wave]
        for a in ads:
            a.options.deferred_metrics = 1
        if len(wave) == 1:
            res = ads[0].excute(wave[0].frames(), nframes=steps)
            per_seq = [res]
        else:
            grp = NS.ReplicaGroup(ads, steps)
            its = [iter(s.frames()) for s in wave]
            for step in range(steps):
                batches = []
                for it in its:
                    b = next(it, None)
                    if b is not None:
                        dev = ads[0].device
                        b = {k: v.to(dev) if isinstance(v, torch.Tensor) else v for k, v in b.items()}
                    batches.append(b)
                grp.step(batches, step)
            per_seq = grp.flush_metrics()
        for s, res in zip(wave, per_seq):
            mp = np.concatenate([np.ravel(np.asarray(x, np.float64)) for x in res["mpjpe"]]) if len(res["mpjpe"]) else np.zeros(0)
            pa = np.concatenate([np.ravel(np.asarray(x, np.float64)) for x in res["pampjpe"]]) if len(res["pampjpe"]) else np.zeros(0)
            pve_b = np.ravel(np.asarray(res["pve"], np.float64))                      # one value per batch (mean over it)
            if mp.shape[0] != s.nframes:             # a missing frame record must not be papered over
                raise RuntimeError(f"sequence {s.name}: {mp.shape[0]} frame records for {s.nframes} frames")
            n = s.nframes
            pve = np.repeat(pve_b, B)[:n] if pve_b.size else np.zeros(n)
            gi = np.arange(s.first, s.first + n, dtype=np.float64)
            rows.append(np.stack([gi, mp[:n], pa[:n], pve], 1))
        del ads
    local = np.concatenate(rows, 0) if rows else np.zeros((0, 4))
    dev = torch.device("cuda", torch.cuda.current_device()) if (torch.cuda.is_available() and torch.distributed.is_available()
                                                                  and torch.distributed.is_initialized()
                                                                  and torch.distributed.get_backend(group) == "nccl") else torch.device("cpu")
    allr = gather_frame_metrics(torch.from_numpy(local).to(dev).flatten(), group).cpu().numpy().reshape(-1, 4)
    allr = allr[np.argsort(allr[:, 0], kind="stable")]
    return dict(global_index=allr[:, 0].astype(np.int64), mpjpe=allr[:, 1], pampjpe=allr[:, 2], pve=allr[:, 3], owned=owned,
                frames_local=int(local.shape[0]))


def main(options):
    """`python -m dynaboa_amd.benchmark --num_shards N [--shard_rank r] [--seqs_per_gpu S]`; under torchrun the rank / world size come
    from the environment (RANK, WORLD_SIZE, LOCAL_RANK) and the final gather runs over RCCL."""
    import os
    import torch.distributed as dist
    from . import benchmark as DB
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    rank = int(os.environ.get("RANK", options.shard_rank))
    nsh = world if world > 1 else int(options.num_shards)
    dev = torch.device("cuda", torch.cuda.current_device())
    seqs = pw3d_sequences(options, dev)
    if int(options.seqs_per_gpu) > 1:
        # several sequences per launch: the throughput policy unless bit-identity with sequences adapted alone is asked for
        from . import native_step as NS
        NS.set_replica_policy(getattr(options, "replica_policy", "throughput") != "bitexact")
    res = run_sharded(options, seqs, lambda: DB.Adaptor(options, device=dev), nsh, rank, int(options.seqs_per_gpu))
    if rank == 0:
        print(f"frames:{len(res['mpjpe'])} MPJPE:{np.mean(res['mpjpe'])}, PAMPJPE:{np.mean(res['pampjpe'])}, PVE:{np.mean(res['pve'])}")
    return res
