"""GPU: the HBM-resident replay and batch maker (hanabizero_amd/device_replay.py, include/hz_replay.h) against the host
restatements that tests/test_reference_callers.py pins to the reference's own batch workers: ReplayBuffer.ingest_packed and
learner.make_batch (core/replay_buffer.py:92-172, core/reanalyze_worker.py:45-168, 249-304, 374-399)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _selfplay(game, N, moves, stack=4, sims=8, dtype=torch.float16, seed=5):
    from tests.test_selfplay import make
    cfg, eng, actor = make(game, N, sims, stack, dtype, use_graph=False, seed=seed)
    cfg.batch_size = 32
    bufs = []
    for rnd in range(2):  # (two packed buffers: the second is appended behind the first)
        for _ in range(moves):
            actor.step()
        torch.cuda.synchronize()
        got = actor.drain_packed()
        assert got is not None
        bufs.append((got[0].clone(), got[1], got[2]))
    return cfg, eng, actor, bufs


def _both(cfg, bufs, capacity=None):
    from hanabizero_amd.device_replay import DeviceReplay
    from hanabizero_amd.replay import ReplayBuffer
    rb = ReplayBuffer(cfg)
    total = sum(b[2] for b in bufs)
    dr = DeviceReplay(cfg, capacity or total + 64, games_capacity=total + 64)  # (Hanabi-Small games are a few moves long: nearly a frame row per position extra)
    for buf, n, moves in bufs:
        rb.ingest_packed(buf.cpu().numpy(), n, moves)
        dr.ingest_packed(buf, n, moves)
    return rb, dr


@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full-5p"])
def test_device_ingest_equals_host_ingest(game):
    cfg, eng, actor, bufs = _selfplay(game, 48, 30 if game == "Hanabi-Small" else 60)
    rb, dr = _both(cfg, bufs)
    assert dr.get_total_len() == rb.get_total_len() > 100 and dr.episodes_collected() == rb.episodes_collected() == len(rb.buffer)
    D, A = cfg.obs_dim, cfg.action_space_size
    p = f = 0
    frames = np.unpackbits(dr.frames[:dr.fhead].cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :D]
    act, rew, vis = dr.action.cpu().numpy(), dr.reward.cpu().numpy(), dr.visits.cpu().numpy().astype(np.float64)
    legal, val = dr.legal.cpu().numpy(), dr.value.cpu().numpy()
    pt, pT, pr0 = dr.pos_t.cpu().numpy(), dr.pos_T.cpu().numpy(), dr.pos_row0.cpu().numpy()
    for g in rb.buffer:
        T = len(g)
        assert (act[p:p + T] == g.actions).all() and (rew[p:p + T] == g.rewards).all()
        assert np.array_equal(vis[p:p + T] / vis[p:p + T].sum(1, keepdims=True), g.child_visits)
        assert np.array_equal(val[p:p + T].astype(np.float64), g.root_values)
        assert (legal[f:f + T + 1] == g.legal_actions).all()
        assert (frames[f:f + T + 1] == g.obs_history[cfg.stacked_observations - 1:]).all()
        assert (pt[p:p + T] == np.arange(T)).all() and (pT[p:p + T] == T).all() and (pr0[p:p + T] == f).all()
        p, f = p + T, f + T + 1
    assert p == dr.head and f == dr.fhead
    assert np.array_equal(dr.priority[:dr.head].cpu().numpy(), rb.priorities)


@pytest.mark.parametrize("game,value_dtype", [("Hanabi-Small", torch.float32), ("Hanabi-Full-5p", torch.float32)])
def test_device_batch_equals_host_make_batch(game, value_dtype):
    """assemble() on positions sampled by the HOST buffer == learner.make_batch of the same (games, positions): model input
    windows, actions (same random padding), reward / value / policy targets, bit for bit, with the same target model."""
    from hanabizero_amd.learner import GraphedUpdate, make_batch, make_optimizer
    cfg, eng, actor, bufs = _selfplay(game, 48, 30 if game == "Hanabi-Small" else 60)
    rb, dr = _both(cfg, bufs)
    B, U, stack, D = cfg.batch_size, cfg.num_unroll_steps, cfg.stacked_observations, cfg.obs_dim
    for rep in range(3):
        games, pos, idx, w, mt = rb.prepare_batch_context(B, beta=0.4)
        value_host = lambda o: eng.initial(torch.from_numpy(np.ascontiguousarray(o, np.float32)).cuda())[0].float().cpu().numpy()
        (obs, actions, mask, _, weights, _), (t_rew, t_val, t_pol) = make_batch(games, pos, cfg, value_host, weights=w, rng=np.random.RandomState(rep))
        out = GraphedUpdate(cfg.get_uniform_network().cuda(), None, cfg, B)  # (its static input tensors; nothing is captured here)
        inside = dr.assemble(torch.from_numpy(np.asarray(idx, np.int64)).cuda(), lambda wdw: eng.initial(wdw)[0], out,
                             rand_actions=torch.from_numpy(actions).cuda())
        assert np.array_equal(out.obs.cpu().numpy(), obs[:, :stack].astype(np.float32))
        assert np.array_equal(out.action.cpu().numpy(), actions)
        assert np.array_equal(out.target_reward.cpu().numpy(), t_rew[:, :U])
        assert np.array_equal(out.target_value.cpu().numpy().view(np.uint32), t_val.view(np.uint32))
        assert np.array_equal(out.target_policy.cpu().numpy().view(np.uint32), t_pol.view(np.uint32))
        assert np.array_equal(inside[:, :U].cpu().numpy(), mask != 0)
    # the device's own random padding: inside a game the stored actions, past its end anything in [0, A)
    ids = torch.from_numpy(np.asarray(idx, np.int64)).cuda()
    dr.assemble(ids, lambda wdw: eng.initial(wdw)[0], out)
    a2 = out.action.cpu().numpy()
    assert np.array_equal(a2[mask != 0], actions[mask != 0]) and a2.min() >= 0 and a2.max() < cfg.action_space_size


def test_replay_windows_kernel_all_layouts():
    """hz_replay_windows: fp32 rows of stack * D elements (the learner's input), 16-bit rows with slots padded to 8 elements (the
    engines' input: 16-B stores) and unpadded 16-bit rows (element stores), windows that reach in front of a game's first frame,
    and rows marked invalid."""
    from hanabizero_amd._lib import HzError
    from hanabizero_amd.device_replay import DeviceReplay
    cfg, eng, actor, bufs = _selfplay("Hanabi-Full-5p", 24, 50)
    rb, dr = _both(cfg, bufs)
    D, stack = cfg.obs_dim, cfg.stacked_observations
    Dp = (D + 7) // 8 * 8
    n = dr.head
    rng = np.random.RandomState(0)
    phys = torch.from_numpy(rng.randint(0, n, 300)).cuda()
    shift = torch.from_numpy(rng.randint(0, 7, 300).astype(np.int32)).cuda()
    valid = (dr.pos_t[phys] + shift) <= dr.pos_T[phys]
    valid[::7] = False
    frames = np.unpackbits(dr.frames[:dr.fhead].cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :D]
    t = (dr.pos_t[phys] + shift).cpu().numpy()
    r0 = dr.pos_row0[phys].cpu().numpy()
    want = np.zeros((300, stack, D), np.float32)
    for m in range(300):
        if bool(valid[m]):
            for j in range(stack):
                want[m, j] = frames[r0[m] + max(0, t[m] - (stack - 1) + j)]
    for dtype, slot in ((torch.float32, D), (torch.float16, Dp), (torch.bfloat16, Dp), (torch.float16, D), (torch.float32, Dp)):
        out = torch.full((300, stack * slot + 3), 7.0, dtype=dtype, device="cuda")[:, :stack * slot]  # (row stride != row length)
        if slot == Dp and dtype != torch.float32:
            out = torch.full((300, stack * slot + 8), 7.0, dtype=dtype, device="cuda")[:, :stack * slot]
        dr.windows(phys, shift, valid, out, slot_elems=slot)
        got = out.float().cpu().numpy().reshape(300, stack, slot)
        assert np.array_equal(got[:, :, :D], want), (dtype, slot)
        assert (got[:, :, D:] == 0).all()
    with pytest.raises(HzError):
        dr.windows(phys, shift, valid, torch.zeros(300, stack * (D - 1), device="cuda"), slot_elems=D - 1)


def test_eviction_and_compaction_keep_batches_right():
    """A replay smaller than what arrives: remove_to_fit drops the oldest whole games, the arrays are compacted, position ids
    handed out before stay valid (or are dropped with their game), and batches still equal the host's for the surviving games."""
    from hanabizero_amd.device_replay import DeviceReplay
    from hanabizero_amd.learner import GraphedUpdate, make_batch
    from hanabizero_amd.replay import ReplayBuffer
    cfg, eng, actor, bufs = _selfplay("Hanabi-Small", 64, 25)
    for _ in range(3):
        for _ in range(25):
            actor.step()
        torch.cuda.synchronize()
        got = actor.drain_packed()
        bufs.append((got[0].clone(), got[1], got[2]))
    total = sum(b[2] for b in bufs)
    cap = int(total * 0.55)
    dr = DeviceReplay(cfg, cap, transition_top=int(total * 0.4), games_capacity=cap)
    rb = ReplayBuffer(cfg, transition_top=int(total * 0.4))
    early = None
    for k, (buf, n, moves) in enumerate(bufs):
        dr.ingest_packed(buf, n, moves)
        rb.ingest_packed(buf.cpu().numpy(), n, moves)
        if k == 0:
            early = dr.sample(16, 0.4)[0]
    assert dr.origin > 0, "the arrays were never compacted: make the capacity smaller"
    dr.remove_to_fit()
    rb.remove_to_fit()
    assert dr.get_total_len() <= int(total * 0.4) and dr.get_total_len() == rb.get_total_len()
    # the host buffer after the same policy holds the same newest games: position i of the live region == host index i
    B, U = cfg.batch_size, cfg.num_unroll_steps
    games, pos, idx, w, mt = rb.prepare_batch_context(B, beta=0.4)
    value_host = lambda o: eng.initial(torch.from_numpy(np.ascontiguousarray(o, np.float32)).cuda())[0].float().cpu().numpy()
    (obs, actions, mask, _, _, _), (t_rew, t_val, t_pol) = make_batch(games, pos, cfg, value_host, rng=np.random.RandomState(0))
    out = GraphedUpdate(cfg.get_uniform_network().cuda(), None, cfg, B)
    ids = torch.from_numpy(np.asarray(idx, np.int64)).cuda() + dr.tail + dr.origin
    dr.assemble(ids, lambda wdw: eng.initial(wdw)[0], out, rand_actions=torch.from_numpy(actions).cuda())
    assert np.array_equal(out.obs.cpu().numpy(), obs[:, :cfg.stacked_observations].astype(np.float32))
    assert np.array_equal(out.target_value.cpu().numpy().view(np.uint32), t_val.view(np.uint32))
    assert np.array_equal(out.target_policy.cpu().numpy().view(np.uint32), t_pol.view(np.uint32))
    # write-backs: ids of evicted positions change nothing that is live; live ones land on their position
    before = dr.priority[dr.tail:dr.head].clone()
    dr.update_priorities(early, torch.full((16,), 123.0, device="cuda"))
    assert torch.equal(dr.priority[dr.tail:dr.head], before), "a write-back for an evicted position reached a live one"
    dr.update_priorities(ids[:5], torch.arange(5, device="cuda").double() + 2)
    assert dr.priority[(ids[:5] - dr.origin)].tolist() == [2.0, 3.0, 4.0, 5.0, 6.0]


def test_prioritised_sampling_distribution_and_weights():
    from hanabizero_amd.device_replay import DeviceReplay
    cfg, eng, actor, bufs = _selfplay("Hanabi-Small", 32, 20)
    _, dr = _both(cfg, bufs)
    n = dr.get_total_len()
    pr = torch.ones(n, dtype=torch.float64, device="cuda")
    pr[: n // 4] = 16.0  # p ** 0.6 = 5.28: the first quarter should be drawn 5.28 x as often per position
    dr.priority[:n] = pr
    hits = torch.zeros(n, device="cuda")
    for _ in range(400):
        ids, w = dr.sample(8, beta=0.4)
        assert ids.unique().numel() == 8, "sampling is without replacement"
        hits[ids] += 1
        probs = pr ** 0.6 / (pr ** 0.6).sum()
        want = (n * probs[ids]) ** -0.4
        assert torch.allclose(w.double(), want / want.max(), rtol=1e-6)
    ratio = float(hits[: n // 4].mean() / hits[n // 4:].mean())
    assert 4.0 < ratio < 6.5, ratio


def test_policy_re_on_the_device_equals_the_host_context_path():
    """policy_re_inputs + policy_re_device == reanalyze.policy_re_context + prepare_policy_re (reanalyze_worker.py:101-144,
    307-371) for the same positions, noise and tie-break seed."""
    from hanabizero_amd.device_replay import policy_re_device
    from hanabizero_amd.reanalyze import policy_re_context, prepare_policy_re
    cfg, eng, actor, bufs = _selfplay("Hanabi-Small", 48, 30, sims=12)
    rb, dr = _both(cfg, bufs)
    U, A = cfg.num_unroll_steps, cfg.action_space_size
    games, pos, idx, w, mt = rb.prepare_batch_context(16, beta=0.4)
    ctx = policy_re_context(cfg, games, pos, idx)
    rng = np.random.RandomState(3)
    noises = rng.dirichlet([0.3] * A, 16 * (U + 1)).astype(np.float32)
    want = prepare_policy_re(cfg, eng, ctx, noises=noises, tie_seed=4)
    win = torch.empty(16 * (U + 1), cfg.obs_shape, dtype=torch.float32, device="cuda")
    legal, mask = dr.policy_re_inputs(torch.from_numpy(np.asarray(idx, np.int64)).cuda(), win)
    assert np.array_equal(win.cpu().numpy(), np.asarray(ctx[0], np.float32).reshape(16 * (U + 1), -1))
    assert np.array_equal(legal.cpu().numpy(), np.asarray(ctx[6]).astype(np.uint8)) and np.array_equal(mask.cpu().numpy(), np.asarray(ctx[1]) != 0)
    got = policy_re_device(cfg, eng, win, legal, mask, noises=torch.from_numpy(noises).cuda(), tie_seed=4)
    assert np.array_equal(got.cpu().numpy().reshape(16, U + 1, A), want.astype(np.float32))
    assert (mask.cpu().numpy() == 0).any(), "no position of the sample ran past its game's end: the masked branch was not exercised"


def test_learner_pipeline_equals_its_single_stream_schedule():
    """learner.LearnerPipeline (prepare / re-search x 2 / learner streams fed by two host threads, batches k + 1 and k + 2 prepared
    while step k trains, priorities written back three steps later, target model refreshed one interval behind) leaves the same weights, priorities and target model as the SAME sequence of
    operations enqueued on one stream -- where ordering is trivially right: any missing event between the streams shows here."""
    import copy
    from hanabizero_amd.learner import LearnerPipeline
    from hanabizero_amd.model import InferenceEngine
    cfg, eng, actor, bufs = _selfplay("Hanabi-Small", 64, 30, sims=10)
    cfg.batch_size, cfg.target_model_interval, cfg.checkpoint_interval = 32, 4, 5
    results = []
    for pipelined in (True, False):
        _, dr = _both(cfg, bufs)
        torch.manual_seed(1)
        model = copy.deepcopy(eng._net).cuda()
        target = InferenceEngine(copy.deepcopy(eng._net), cfg.value_support.max, dtype=torch.float16, device="cuda")
        calls = []
        pipe = LearnerPipeline(cfg, dr, model, target, reanalyze_share=0.5, on_checkpoint=lambda step, ev: calls.append(step), seed=3,
                               host_thread=pipelined)
        if not pipelined:
            pipe.prep = pipe.learn = torch.cuda.current_stream()
            pipe.side = [torch.cuda.current_stream()] * len(pipe.side)
        for _ in range(11):
            pipe.step()
        pipe.flush()
        torch.cuda.synchronize()
        assert pipe.steps == 11 and calls == [5, 10]
        losses = pipe.losses()
        pipe.close()
        assert all(np.isfinite(x) for x in losses)
        results.append(([p.detach().clone() for p in model.parameters()], dr.priority[:dr.head].clone(),
                        [t.clone() for t in target._dev.values()], losses))
        assert not torch.equal(results[-1][1], torch.ones_like(results[-1][1])), "no priority was written back"
    for a, b in zip(results[0][0], results[1][0]):
        assert torch.equal(a, b), "the pipelined learner's weights differ from the single-stream schedule's"
    assert torch.equal(results[0][1], results[1][1])
    for a, b in zip(results[0][2], results[1][2]):
        assert torch.equal(a, b)
    assert results[0][3] == results[1][3]


@pytest.mark.parametrize("ranks", [1, 2])
def test_config5_loop_roles_on_one_gpu(ranks):
    """tools/loop_bench.py, the configs[4] loop as DESIGN.md section 5 lays it out -- at one rank (learner + actor in one process,
    three streams) and as a two-rank rehearsal on the test box's one GPU over gloo (rank 0 learns, rank 1 acts: packed games
    gathered into the device replay, weights broadcast back and taken over in place) -- small enough for a test: every role's
    hand-offs happen at least once and the one JSON line adds up."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "tools/loop_bench.py", "--gpus", str(ranks), "--game", "Hanabi-Full-5p", "--envs", "256", "--rounds", "3",
           "--warm-rounds", "8", "--flush-every", "10", "--ratio", "0.004", "--batch-size", "64", "--simulations", "20",
           "--checkpoint-interval", "8", "--target-interval", "6", "--replay-capacity", "200000"]
    if ranks > 1:
        cmd += ["--backend", "gloo", "--share-device"]
    p = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["roles"]["acting_ranks"] == 1 and d["roles"]["learner_acts"] == (ranks == 1)
    moves = 3 * 10 * 256
    assert d["learner_steps"] == int(moves * 0.004) >= 28 and d["replay_ratio_achieved"] == pytest.approx(0.004, rel=0.05)
    assert d["selfplay_moves_per_s"] == pytest.approx(moves / d["wall_s"], rel=1e-6)
    assert d["games_ingested"] > 50 and d["replay_positions"] > 1000 and d["weight_handovers_in_run"] >= 2 and d["weight_handover_first_ms"] > 0
    assert d["loss_last"] is not None and np.isfinite(d["loss_last"]) and d["weight_handover_ms"] > 0


def test_streams_on_distinct_queues_run_side_by_side():
    """learner.streams_on_distinct_queues: the streams it returns overlap pairwise (a chain of small dependent GEMMs on each of two
    takes about what one takes), which two arbitrary streams of the process need not (the GPU has 4 hardware queues)."""
    from hanabizero_amd.learner import streams_on_distinct_queues
    dev = torch.device("cuda", torch.cuda.current_device())
    streams, one, pairs = streams_on_distinct_queues(dev, 4)
    assert len(streams) == 4 and len({s.cuda_stream for s in streams}) == 4 and one > 0
    accepted = [p for p in pairs if max(p) < 1.4 * one]
    print("one chain %.2f ms; pairs %s" % (one, [[round(x, 2) for x in p] for p in pairs]))
    # (on this pool: three candidates join the first stream -- four hardware queues; asserted loosely: a timing test must not be
    # what turns the suite red on a box whose runtime hands out queues differently)
    assert len(accepted) >= 1, (one, pairs)


@pytest.mark.parametrize("n", [300_001, 1_048_576, 5_000_003])
def test_chunked_top_k_of_the_sampler_is_exact(n):
    """device_replay._topk_indices (the k chunks with the largest maxima hold the k largest keys: what keeps prioritised sampling at
    a fraction of a millisecond on a replay of the reference's 25 M positions) == torch.topk, as a set -- with a cluster of
    maximum-priority positions (freshly ingested games) and winners in the ragged tail behind the last whole chunk."""
    from hanabizero_amd.device_replay import _topk_indices
    g = torch.Generator(device="cuda").manual_seed(n)
    p = (torch.rand(n, device="cuda", dtype=torch.float64, generator=g) + 0.01) ** 0.6
    p[n // 2:n // 2 + n // 80] = p.max() * 1.5
    p[-40:] = p.max() * 1e4      # (so heavy that the ragged tail is sure to hold winners)
    keys = p / torch.empty_like(p).exponential_(1.0, generator=g)
    got = torch.sort(_topk_indices(keys, 256)).values
    want = torch.sort(torch.topk(keys, 256, sorted=False).indices).values
    assert torch.equal(got, want)
    assert bool((want >= n - 40).any()), "the case is meant to have winners in the tail"
