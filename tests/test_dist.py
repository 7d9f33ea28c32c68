"""CPU, world_size 2, gloo: the N>1 exchange steps (record gather to the replay owner, weight broadcast)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_rec(n, seed, T=6, A=11, W=7):
    rng = np.random.RandomState(seed)
    return dict(action=rng.randint(0, A, (n, T)).astype(np.int8), reward=rng.randint(-2, 3, (n, T)).astype(np.int8),
                value=rng.rand(n, T).astype(np.float32), visits=rng.randint(0, 9, (n, T, A)).astype(np.int16),
                legal=rng.randint(0, 2, (n, T + 1, A)).astype(np.uint8), obs=rng.randint(-2**31, 2**31 - 1, (n, T + 1, W)).astype(np.int32),
                meta=np.concatenate([rng.randint(1, T + 1, (n, 1)), rng.randint(0, 100, (n, 3))], 1).astype(np.int32))


def _worker(rank, world, port, counts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hanabizero_amd.dist import broadcast_weights, gather_records
    ok = True
    for rnd, cnts in enumerate(counts):
        rec = _fake_rec(cnts[rank], 100 * rnd + rank) if cnts[rank] else None
        got = gather_records(rec, dst=0)
        if rank == 0:
            want = [_fake_rec(c, 100 * rnd + r) for r, c in enumerate(cnts) if c]
            if not want:
                ok &= got is None
            else:
                for k in want[0]:
                    ok &= bool((got[k] == np.concatenate([w[k] for w in want], 0)).all()) and got[k].dtype == want[0][k].dtype
        else:
            ok &= got is None
    # ranks trim their records to their own longest finished game (SelfPlayActor.drain): different time extents
    Ts = (4, 9)
    rec = _fake_rec(3, 7 + rank, T=Ts[rank])
    got = gather_records(rec, dst=0)
    if rank == 0:
        ok &= got["action"].shape == (6, 9) and got["obs"].shape[:2] == (6, 10) and got["meta"].shape == (6, 4)
        for r in range(2):
            w = _fake_rec(3, 7 + r, T=Ts[r])
            for k in w:
                sl = (slice(3 * r, 3 * r + 3),) + tuple(slice(0, d) for d in w[k].shape[1:])
                ok &= bool((got[k][sl] == w[k]).all())
        ok &= bool((got["action"][:3, 4:] == 0).all())  # rank 0's rows are zero-padded up to rank 1's extent
    # the device-resident form: packed byte buffers (here CPU tensors over gloo), different (n, tmax) per rank, one empty
    from hanabizero_amd.dist import gather_packed
    from hanabizero_amd.selfplay import pack_records, unpack_packed, unpack_record
    A, W = 11, 7

    def pack(rec_):
        buf, n_, moves_ = pack_records(rec_, A, W)
        return torch.from_numpy(buf), n_, moves_
    for rnd, ns in enumerate([(3, 2), (0, 4), (0, 0)]):
        mine = _fake_rec(ns[rank], 50 + 10 * rnd + rank, T=5 + 3 * rank, A=A, W=W) if ns[rank] else None
        got = gather_packed(None if mine is None else pack(mine), A, W, dst=0)
        if rank == 0:
            want = [(r, _fake_rec(c, 50 + 10 * rnd + r, T=5 + 3 * r, A=A, W=W)) for r, c in enumerate(ns) if c]
            ok &= len(got) == len(want)
            for (buf, n_, moves_), (r, w) in zip(got, want):
                view = unpack_packed(buf, n_, moves_, A, W)
                ok &= n_ == ns[r] and moves_ == int(w["meta"][:, 0].sum())
                for i in range(n_):
                    a, b = unpack_record(view, i), unpack_record(w, i)
                    for k in a:
                        ok &= bool(np.array_equal(np.asarray(a[k]), np.asarray(b[k])))
        else:
            ok &= got is None
    # a state_dict as a net has it: several fp32 tensors of different shapes + int64 counters -> one flat broadcast per dtype
    sd = {"b": torch.full((3,), float(rank)), "a": torch.arange(4.0) * (rank + 1), "w": torch.arange(6.0).reshape(2, 3) + rank,
          "bn.num_batches_tracked": torch.tensor(7 + rank), "steps": torch.tensor([1, 2]) * (rank + 1)}
    out = broadcast_weights(sd, src=0)
    ok &= bool((out["b"] == 0).all()) and bool((out["a"] == torch.arange(4.0)).all())
    ok &= out["w"].shape == (2, 3) and bool((out["w"] == torch.arange(6.0).reshape(2, 3)).all())
    ok &= out["bn.num_batches_tracked"].dtype == torch.int64 and out["bn.num_batches_tracked"].shape == () and int(out["bn.num_batches_tracked"]) == 7
    ok &= bool((out["steps"] == torch.tensor([1, 2])).all()) and set(out) == set(sd)
    q.put((rank, ok))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_gather_and_broadcast_world2_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    counts = [(3, 5), (0, 2), (4, 0), (0, 0)]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _worker_packed(rank, world, port, q):
    """gather_packed's point-to-point exchange with more than one sender: ranks with games send their exact bytes to rank 0,
    ranks without send nothing, rank 0's own games take no transfer; three rounds, buffers reused."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hanabizero_amd.dist import gather_packed
    from hanabizero_amd.selfplay import pack_records, unpack_packed, unpack_record
    A, W = 20, 25
    ok = True
    for rnd, ns in enumerate([(2, 0, 5, 1), (0, 3, 0, 4), (1, 1, 1, 1), (0, 0, 0, 0)]):
        n = ns[rank]
        mine = _fake_rec(n, 900 + 10 * rnd + rank, T=4 + 2 * rank, A=A, W=W) if n else None
        packed = None
        if mine is not None:
            buf, n_, moves_ = pack_records(mine, A, W)
            packed = (torch.from_numpy(buf), n_, moves_)
        got = gather_packed(packed, A, W, dst=0)
        if rank == 0:
            want = [(r, _fake_rec(c, 900 + 10 * rnd + r, T=4 + 2 * r, A=A, W=W)) for r, c in enumerate(ns[:world]) if c]
            ok &= len(got) == len(want)
            for (buf, n_, moves_), (r, w) in zip(got, want):
                view = unpack_packed(buf, n_, moves_, A, W)
                ok &= n_ == ns[r] and moves_ == int(w["meta"][:, 0].sum())
                for i in range(n_):
                    a, b = unpack_record(view, i), unpack_record(w, i)
                    ok &= all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in a)
        else:
            ok &= got is None
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_packed_world4_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_packed, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(4)]


def _worker_subgroup(rank, world, port, q):
    """r04 (advisor): `dst` / `src` are ranks OF THE GROUP; the point-to-point calls underneath want global ranks.  A subgroup
    {1, 3} of four ranks gathers to its member 1 (= global rank 3) in the device form (to_host=False: torch buffers, as the
    HBM-resident replay ingests them) and broadcasts weights from it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hanabizero_amd.dist import broadcast_weights, gather_packed
    from hanabizero_amd.selfplay import pack_records, unpack_packed, unpack_record
    sub = dist.new_group([1, 3])
    ok = True
    if rank in (1, 3):
        A, W = 11, 7
        for rnd in range(2):
            mine = _fake_rec(2 + rank + rnd, 40 + 10 * rnd + rank, T=5, A=A, W=W)
            buf, n_, moves_ = pack_records(mine, A, W)
            got = gather_packed((torch.from_numpy(buf), n_, moves_), A, W, dst=1, group=sub, to_host=False)
            if rank == 3:
                ok &= len(got) == 2 and all(isinstance(b, torch.Tensor) for b, _, _ in got)
                for (b, n2, m2), r in zip(got, (1, 3)):   # (group-rank order: global ranks 1, 3)
                    w = _fake_rec(2 + r + rnd, 40 + 10 * rnd + r, T=5, A=A, W=W)
                    view = unpack_packed(b.numpy(), n2, m2, A, W)
                    ok &= n2 == 2 + r + rnd
                    for i in range(n2):
                        a, c = unpack_record(view, i), unpack_record(w, i)
                        ok &= all(np.array_equal(np.asarray(a[k]), np.asarray(c[k])) for k in a)
            else:
                ok &= got is None
        out = broadcast_weights({"w": torch.full((4,), float(rank))}, src=1, group=sub)
        ok &= bool((out["w"] == 3.0).all())
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gather_packed_subgroup_ranks_and_device_form_gloo():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_subgroup, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(4)]


@pytest.mark.gpu
def test_collectives_over_rccl_one_rank():
    """The same three exchange steps through the "nccl" (= RCCL) backend with device tensors: one rank on the one GPU of
    the test box (RCCL refuses two ranks on one device), with the single-rank short cut switched off so that
    all_gather / gather / broadcast really go through the RCCL communicator -- tensor placement, dtypes and call
    signatures are what differs from gloo."""
    import hanabizero_amd.dist as hd
    from hanabizero_amd.selfplay import pack_records, unpack_packed, unpack_record
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    hd._SHORT_CIRCUIT = False
    try:
        dist.barrier()
        rec = _fake_rec(5, 3, T=7)
        got = hd.gather_records(rec, dst=0)
        for k in rec:
            assert got[k].dtype == rec[k].dtype and (got[k] == rec[k]).all(), k
        assert hd.gather_records(None, dst=0) is None
        A, W = 11, 7
        buf, n_, moves_ = pack_records(rec, A, W)
        out = hd.gather_packed((torch.from_numpy(buf).to(device), n_, moves_), A, W, dst=0)
        assert len(out) == 1 and out[0][1:] == (n_, moves_)
        view = unpack_packed(out[0][0], n_, moves_, A, W)
        for i in range(n_):
            a, b = unpack_record(view, i), unpack_record(rec, i)
            for k in a:
                assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (i, k)
        assert hd.gather_packed(None, A, W, dst=0) == []
        # r04: the games stay on the device for a replay that lives there (tools/loop_bench.py: gather -> DeviceReplay.ingest_packed)
        dev_buf = torch.from_numpy(buf).to(device)
        out = hd.gather_packed((dev_buf, n_, moves_), A, W, dst=0, to_host=False)
        assert len(out) == 1 and out[0][0].is_cuda and out[0][1:] == (n_, moves_) and torch.equal(out[0][0][:dev_buf.numel()], dev_buf)
        assert hd.gather_packed(None, A, W, dst=0, to_host=False) == []
        # ... and subgroup ranks are translated (group rank 0 of a one-member subgroup is global rank 0 here: the call shape is what is exercised)
        sub = dist.new_group([0])
        out = hd.gather_packed((dev_buf, n_, moves_), A, W, dst=0, group=sub, to_host=False)
        assert len(out) == 1 and out[0][1:] == (n_, moves_)
        sd = {"w": torch.arange(6.0).reshape(2, 3), "b": torch.ones(3, dtype=torch.bfloat16)}
        bw = hd.broadcast_weights(sd, src=0)
        assert bw["w"].is_cuda and (bw["w"].cpu() == sd["w"]).all() and bw["b"].dtype == torch.bfloat16
        t = torch.tensor([1.5], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks of the elapsed time
        assert float(t.item()) == 1.5
    finally:
        hd._SHORT_CIRCUIT = True
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_two_ranks_on_one_gpu_over_gloo(launcher):
    """bench.py's N > 1 path both ways it can be started -- under python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2 ... , and as plain `python bench.py --gpus 2 ...` (no WORLD_SIZE in the environment: bench.py starts its own two
    ranks before it touches a GPU) -- rehearsed on the test box's one GPU: both ranks use cuda:0, the collectives go over gloo.
    Checks the one JSON line rank 0 prints: two ranks' worth of moves, finished games gathered from BOTH ranks (global env ids
    from each rank's disjoint range), the asynchronous drain in use."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port())] if launcher == "torchrun" else [sys.executable]
    cmd = head + ["bench.py", "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "24",
                  "--warmup", "2", "--workload", "small4096", "--flush-every", "8", "--no-cpu-baseline", "--no-roofline", "--no-also",
                  "--check-env-ids"]
    p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    N = 4096
    assert d["n_gpus"] == 2 and d["steps"] == 24 and d["scaling"] == "weak" and d["metric"] == "selfplay_moves_per_sec"
    assert d["value"] == pytest.approx(2 * N * 24 / (d["ms_per_step"] * 24e-3), rel=1e-6)
    cfg = d["config"]
    assert cfg["drain"].startswith("asynchronous") and cfg["games_finished"] > N and cfg["record_bytes_gathered"] > 0
    lo, hi, distinct = cfg["env_id_min_max_distinct"]
    assert 0 <= lo < N <= hi < 2 * N and distinct > N  # games of rank 0's envs [0, N) and of rank 1's [N, 2N)
