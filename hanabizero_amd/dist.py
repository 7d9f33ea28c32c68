"""hanabizero_amd.dist -- the two exchange steps of the sharded self-play job.

The reference moves data between its Ray actors through the plasma object store: finished games
(``replay_buffer.save_pools.remote``, /root/reference/core/selfplay_worker.py:75-78) and weights
(``shared_storage.get_weights.remote``, :181).  Here one actor process owns one GPU; trees and envs never talk to
each other inside a move, so the only collectives are
  * ``gather_records``   finished-game packed records -> the replay owner (rank 0), over RCCL/xGMI when the process
                         group is "nccl" (device tensors, direct peer->root transfers), over gloo on CPU in tests;
  * ``broadcast_weights`` learner rank -> every actor, every checkpoint_interval learner steps.
Both are latency-bound (a few hundred KB per lock-step per GPU); neither is inside the timed hot loop of a move.
"""
import numpy as np
import torch
import torch.distributed as dist

_FIELDS = ("action", "reward", "value", "visits", "legal", "obs", "meta")


def _to_dev(a, device):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device) if device.type == "cuda" else t


def gather_records(rec, dst=0, group=None, device=None):
    """rec: dict of numpy arrays with a leading games axis (SelfPlayActor.drain()) or None.
    Returns on `dst` the concatenation over ranks (rank order), elsewhere None.  Collective: every rank must call."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rec
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    device = torch.device(device) if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    n = 0 if rec is None else int(rec["meta"].shape[0])
    counts = torch.tensor([n], dtype=torch.int64, device=device)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = [int(c.item()) for c in all_counts]
    mx = max(all_counts)
    if mx == 0:
        return None
    # every rank trims its records to its own longest finished game (SelfPlayActor.drain): agree on the element-wise
    # maximum of the tail shapes (dtypes are the same everywhere) and pad to it
    spec = None if rec is None else {k: (tuple(rec[k].shape[1:]), rec[k].dtype.str) for k in _FIELDS}
    specs = [None] * world
    dist.all_gather_object(specs, spec, group=group)
    have = [s for s in specs if s is not None]
    spec = {k: (tuple(max(s[k][0][d] for s in have) for d in range(len(have[0][k][0]))), have[0][k][1]) for k in _FIELDS}
    out = {} if rank == dst else None
    for k in _FIELDS:
        shape, dt = spec[k]
        pad = np.zeros((mx,) + tuple(shape), dtype=np.dtype(dt))
        if n:
            pad[(slice(0, n),) + tuple(slice(0, d) for d in rec[k].shape[1:])] = rec[k]
        t = _to_dev(pad.view(np.uint8).reshape(mx, -1), device)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, bufs, dst=dst, group=group)
        if rank == dst:
            parts = [b.cpu().numpy()[:c].reshape(-1).view(np.dtype(dt)).reshape((c,) + tuple(shape))
                     for b, c in zip(bufs, all_counts) if c]
            out[k] = np.concatenate(parts, 0)
    return out


def broadcast_weights(state_dict, src=0, group=None, device=None):
    """Broadcast a model state_dict from `src` in place (tensors are moved to `device` for nccl)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return state_dict
    backend = dist.get_backend(group)
    device = torch.device(device) if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    out = {}
    for k in sorted(state_dict.keys()):
        t = state_dict[k].detach().to(device).contiguous()
        dist.broadcast(t, src=src, group=group)
        out[k] = t
    return out
