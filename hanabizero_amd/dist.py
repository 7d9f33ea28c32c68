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

_SHORT_CIRCUIT = True  # tests clear this to drive the collectives themselves through a one-rank "nccl" group on one GPU


def _alone(group):
    return not dist.is_initialized() or (_SHORT_CIRCUIT and dist.get_world_size(group) == 1)


_FIELDS = ("action", "reward", "value", "visits", "legal", "obs", "meta")


def _to_dev(a, device):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device) if device.type == "cuda" else t


_DTYPES = dict(action=np.int8, reward=np.int8, value=np.float32, visits=np.int16, legal=np.uint8, obs=np.int32,
               meta=np.int32)  # the packed record format of SelfPlayActor.drain


def gather_records(rec, dst=0, group=None, device=None):
    """rec: dict of numpy arrays with a leading games axis (SelfPlayActor.drain()) or None.
    Returns on `dst` the concatenation over ranks (rank order), elsewhere None.  Collective: every rank must call.
    Two collectives per call whatever the number of fields: an all_gather of the ranks' array shapes (every rank trims its
    records to its own longest finished game, so the time extents differ) and one gather of a byte buffer holding all
    seven arrays padded to the element-wise maximum shape."""
    if _alone(group):
        return rec
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    backend = dist.get_backend(group)
    device = torch.device(device) if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    n = 0 if rec is None else int(rec["meta"].shape[0])
    shp = np.ones(1 + 3 * len(_FIELDS), np.int64)
    shp[0] = n
    if n:
        for i, k in enumerate(_FIELDS):
            assert rec[k].dtype == _DTYPES[k] and rec[k].ndim <= 4, (k, rec[k].dtype, rec[k].shape)
            tail = rec[k].shape[1:]
            shp[1 + 3 * i:1 + 3 * i + len(tail)] = tail
    mine = torch.from_numpy(shp).to(device)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    every = torch.stack(every).cpu().numpy()
    counts = [int(c) for c in every[:, 0]]
    mx = max(counts)
    if mx == 0:
        return None
    have = every[every[:, 0] > 0]
    tails = {k: tuple(int(x) for x in have[:, 1 + 3 * i:4 + 3 * i].max(0)) for i, k in enumerate(_FIELDS)}
    sizes = {k: mx * int(np.prod(tails[k])) * np.dtype(_DTYPES[k]).itemsize for k in _FIELDS}
    buf = np.zeros(sum(sizes.values()), np.uint8)
    off = 0
    for k in _FIELDS:
        if n:
            view = buf[off:off + sizes[k]].view(_DTYPES[k]).reshape((mx,) + tails[k])
            src = rec[k].reshape(rec[k].shape + (1,) * (4 - rec[k].ndim))
            view[(slice(0, n),) + tuple(slice(0, d) for d in src.shape[1:])] = src
        off += sizes[k]
    t = _to_dev(buf, device)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, bufs, dst=dst if group is None else dist.get_global_rank(group, dst), group=group)  # (`dst`: a rank of `group`)
    if rank != dst:
        return None
    out = {k: [] for k in _FIELDS}
    for r, c in enumerate(counts):
        if not c:
            continue
        b = bufs[r].cpu().numpy()
        off = 0
        for i, k in enumerate(_FIELDS):
            full = b[off:off + sizes[k]].view(_DTYPES[k]).reshape((mx,) + tails[k])
            out[k].append(full[:c])
            off += sizes[k]
    # drop the padding axes again (every field has its own rank: meta / action 2-D, visits 3-D, ...)
    ranks = dict(action=2, reward=2, value=2, visits=3, legal=3, obs=3, meta=2)
    res = {k: np.concatenate(out[k], 0) for k in _FIELDS}
    return {k: v.reshape(v.shape[:ranks[k]]) for k, v in res.items()}


_pinned = {}


def _pinned_bytes(key, nbytes):
    buf = _pinned.get(key)
    if buf is None or buf.numel() < nbytes:  # grows geometrically: pinning host memory costs milliseconds
        size = max(int(nbytes), 1, 2 * (buf.numel() if buf is not None else 0))
        buf = torch.zeros(size, dtype=torch.uint8, pin_memory=torch.cuda.is_available())  # (zeros: touch every page now)
        _pinned[key] = buf
    return buf[:nbytes]


def reserve_landing(nbytes, ranks=1):
    """Pin the replay owner's landing buffers (one per sending rank) ahead of time, so that no drain has to."""
    for r in range(ranks):
        _pinned_bytes(("r", r), int(nbytes))


_staging = {}


def _staging_bytes(key, nbytes, device):
    """A reusable byte buffer on `device` (send side: one; receive side: one per sending rank), grown geometrically: a drain
    allocates nothing once the run has seen its largest interval."""
    buf = _staging.get(key)
    if buf is None or buf.numel() < nbytes or buf.device != device:
        size = max(int(nbytes), 1 << 16, 2 * (buf.numel() if buf is not None and buf.device == device else 0))
        buf = _staging[key] = torch.empty(size, dtype=torch.uint8, device=device)
    return buf[:nbytes]


last_gather = {"exchange_s": 0.0, "landing_s": 0.0}  # host seconds the last gather_packed spent in its two halves (bench.py adds them up)


def gather_packed(packed, A, W, dst=0, group=None, to_host=True):
    """The device-resident form of gather_records: `packed` = SelfPlayActor.drain_packed() (uint8 device tensor, n games, their total moves)
    or None.  One all_gather of (n, moves), then every rank with games sends its byte buffer -- its exact size, no padding --
    straight to `dst` (point-to-point, batched: device to device over xGMI's direct peer links under "nccl"; the sending
    ranks never copy their records to the host; `dst`'s own games take no collective at all).  Send and receive staging
    buffers are kept between calls.  Returns on `dst` a list of (host uint8 numpy buffer in pinned memory, n, moves) per rank
    with games -- `unpack_packed` views them without a copy; the buffers are reused by the next call -- elsewhere None.
    Works without a process group (world 1).  Everything is enqueued on the CALLER's current stream and only that stream is
    waited for, so called under ``with torch.cuda.stream(actor.drain_stream)`` the gather overlaps the lock-steps queued on
    the main stream.  `dst` is a rank OF `group` (translated to the global rank the point-to-point calls want).  Stream contract
    for `packed[0]` on the sending ranks: under "nccl" the send is only ordered on the caller's stream when this returns, so the
    buffer may be rewritten by work enqueued on that same stream (what SelfPlayActor.drain_end's next pack is) and by nothing else.
    to_host=False: `dst` gets the buffers as they arrived -- torch uint8 tensors on the device under "nccl" (host tensors in a
    gloo rehearsal) -- for a replay that lives in HBM (hanabizero_amd.device_replay.DeviceReplay.ingest_packed): nothing is copied
    to the host at all; the receive buffers are reused by the next call, so ingest them on the calling stream before it."""
    import time
    from .selfplay import packed_layout
    n, moves = (0, 0) if packed is None else (int(packed[1]), int(packed[2]))
    t0 = time.perf_counter()
    if _alone(group):
        last_gather["exchange_s"] = 0.0
        if not n:
            last_gather["landing_s"] = 0.0
            return []
        if not to_host:
            last_gather["landing_s"] = 0.0
            return [(packed[0], n, moves)]
        host = _pinned_bytes(("r", 0), packed[0].numel())
        host.copy_(packed[0], non_blocking=True)
        if packed[0].is_cuda:  # (this stream only: lock-steps queued on other streams keep running)
            torch.cuda.current_stream(packed[0].device).synchronize()
        last_gather["landing_s"] = time.perf_counter() - t0
        return [(host.numpy(), n, moves)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    peer = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))  # P2POp's `peer` is a global rank
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor([n, moves], dtype=torch.int64, device=device)
    every = _staging_bytes(("counts", world), 16 * world, device).view(torch.int64).view(world, 2)
    dist.all_gather_into_tensor(every, mine, group=group) if backend == "nccl" else dist.all_gather(list(every.unbind(0)), mine, group=group)
    every = every.cpu().numpy().copy()
    sizes = [packed_layout(int(a), int(b), A, W)[1] if a else 0 for a, b in every]
    ops, recv = [], {}
    if rank == dst:
        for r in range(world):
            if sizes[r] and r != rank:
                recv[r] = _staging_bytes(("recv", r), sizes[r], device)
                ops.append(dist.P2POp(dist.irecv, recv[r], peer(r), group=group))
        if sizes[rank]:
            recv[rank] = packed[0][:sizes[rank]]
    elif n:
        send = packed[0][:sizes[rank]]
        if send.device != device:  # (a gloo rehearsal of device-resident records: through the host)
            send = _staging_bytes(("send",), sizes[rank], device).copy_(send)
        ops.append(dist.P2POp(dist.isend, send.contiguous(), peer(dst), group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    t1 = time.perf_counter()
    last_gather["exchange_s"] = t1 - t0
    if rank != dst:
        last_gather["landing_s"] = 0.0
        return None
    if not to_host:
        last_gather["landing_s"] = 0.0
        return [(recv[r], int(every[r, 0]), int(every[r, 1])) for r in range(world) if sizes[r]]
    out = []
    for r in range(world):
        if sizes[r]:
            host = _pinned_bytes(("r", r), sizes[r])
            host.copy_(recv[r], non_blocking=True)
            out.append((host, int(every[r, 0]), int(every[r, 1])))
    if torch.cuda.is_available() and any(b.is_cuda for b in recv.values()):  # (this stream only: lock-steps queued on other streams keep running)
        torch.cuda.current_stream().synchronize()
    last_gather["landing_s"] = time.perf_counter() - t1
    return [(h.numpy(), a, b) for h, a, b in out]


def broadcast_weights(state_dict, src=0, group=None, device=None):
    """Broadcast a model state_dict from `src`: ONE flat buffer and one broadcast per dtype present (fp32 parameters and
    running statistics; the int64 num_batches_tracked counters) -- two collectives for the ~60 tensors / 6 MB of a Hanabi net
    instead of one per tensor.  Returns {name: tensor on `device`} (views of the flat buffers, in the state_dict's shapes)."""
    if _alone(group):
        return state_dict
    backend = dist.get_backend(group)
    device = torch.device(device) if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu"))
    keys = sorted(state_dict.keys())
    out = {}
    for dt in sorted({state_dict[k].dtype for k in keys}, key=str):
        ks = [k for k in keys if state_dict[k].dtype == dt]
        flat = torch.cat([state_dict[k].detach().reshape(-1).to(device) for k in ks]) if ks else None
        dist.broadcast(flat, src=src if group is None else dist.get_global_rank(group, src), group=group)  # (`src`: a rank of `group`)
        off = 0
        for k in ks:
            m = state_dict[k].numel()
            out[k] = flat[off:off + m].view(state_dict[k].shape)
            off += m
    return out
