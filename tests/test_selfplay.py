"""GPU: the device-resident self-play actor against an independent replay of the same moves through the ORACLE
tree + ORACLE env + a numpy restatement of select_action (core/utils.py:280-295), sharing only the nets."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


from tests.restate import ref_select_action  # noqa: E402  (checked against the reference: tests/test_reference_callers.py)


def make(game, N, sims, stack, dtype, use_graph, seed=3, peaked=False, fused_tail=True, max_moves=None, fused=None):
    from hanabizero_amd.config import make_config
    from hanabizero_amd.model import InferenceEngine
    from hanabizero_amd.selfplay import SelfPlayActor
    from tests.netgold import fill_state_dict
    cfg = make_config(game, simulations=sims, stack=stack, p_mcts_num=N)
    if max_moves is not None:  # (trajectory rows of another length: the row copies' alignment cases)
        cfg.max_moves = cfg.test_max_moves = max_moves
    net = cfg.get_uniform_network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
    if peaked is True:  # a policy head that all but always proposes one action and constant values / rewards: every simulation
        with torch.no_grad():  # extends one chain (deep paths)
            net._prediction_actor[-1].weight.zero_()   # (the same logits at every node)
            net._prediction_actor[-1].bias[5] += 40.0  # (play card 0: legal in every position)
            for head in (net._prediction_value, net._dynamics_reward):
                head[-1].weight.zero_()
                head[-1].bias.zero_()
    if peaked == "sharp":  # a concentrated policy that still depends on the state (bench.py's deep_paths net): long paths that
        with torch.no_grad():  # repeat from one simulation to the next and now and then branch off -- replays that end early
            for head in (net._prediction_value, net._dynamics_reward, net._prediction_actor):
                head[-1].weight.normal_(0, 0.1, generator=torch.Generator().manual_seed(1))
                head[-1].bias.normal_(0, 0.1, generator=torch.Generator().manual_seed(2))
            net._prediction_actor[-1].weight.mul_(40.0)
            net._prediction_actor[-1].bias.mul_(40.0)
    net.eval()
    eng = InferenceEngine(net, cfg.value_support.max, dtype=dtype, device="cuda", fused=fused)
    return cfg, eng, SelfPlayActor(cfg, eng, N, seed=seed, use_graph=use_graph, fused_tail=fused_tail)


@pytest.mark.parametrize("game,N,sims,stack,steps", [("Hanabi-Small", 16, 10, 2, 40), ("Hanabi-Full", 8, 12, 4, 25)])
def test_actor_matches_oracle_replay(game, N, sims, stack, steps):
    from oracle.cport import OracleEnv, OracleTree
    from hanabizero_amd.game import GameHistory
    from hanabizero_amd.selfplay import unpack_record
    cfg, eng, actor = make(game, N, sims, stack, torch.float32, use_graph=False)
    A, D = cfg.action_space_size, cfg.obs_dim
    seeds = 3 + np.arange(N)
    oenv = OracleEnv(game, seeds)
    oenv.reset()
    obs, legal = oenv.observe()
    windows = [[obs[i].copy() for _ in range(stack)] for i in range(N)]
    hist = [dict(obs=[obs[i].copy()], legal=[legal[i].copy()], action=[], reward=[], visits=[], value=[]) for i in range(N)]
    finished = []
    for step in range(steps):
        actor._draw()
        noise, uni = actor.noise.cpu().numpy(), actor.uniform.cpu().numpy()
        stack_in = np.stack([np.concatenate(w) for w in windows]).astype(np.float32)
        assert (actor.stack_buf[:, :, :D].reshape(N, -1).float().cpu().numpy() == stack_in).all(), step
        assert (actor.legal.cpu().numpy() == legal).all()
        actor._step_body(draw=False)  # (the draws above are this move's)
        # ---- oracle replay of the same move
        # (the same first-layer GEMM as the actor: observation slots padded to actor.Dp columns)
        win = np.zeros((N, stack, actor.Dp), np.float32)
        win[:, :, :D] = stack_in.reshape(N, stack, D)
        v0, l0, h0 = eng.initial(torch.from_numpy(win.reshape(N, -1)).cuda(), padded=actor.Dp != D)
        tree = OracleTree(N, A, sims, seed=3, value_delta_max=cfg.value_delta_max)
        tree.prepare(cfg.root_exploration_fraction, noise, np.zeros(N, np.float32), l0.cpu().numpy(), legal)
        pool = [h0]
        for sim in range(sims - 1):
            ix, iy, la = tree.traverse(sim, cfg.pb_c_base, cfg.pb_c_init, cfg.discount)
            hid = torch.stack([pool[x][y] for x, y in zip(ix, iy)])
            net_in = torch.zeros(N, eng.H + eng.onehot_cols, dtype=eng.dtype, device="cuda")
            net_in[:, :eng.H] = hid
            net_in[torch.arange(N), eng.H + torch.from_numpy(la).long()] = 1
            h = torch.empty(N, eng.H, dtype=eng.dtype, device="cuda")
            r_log, v_log, p_log = eng.recurrent_heads(net_in, h)  # the same GEMM sequence the actor's search runs
            pool.append(h)
            r, v = eng.support_to_scalar(r_log), eng.support_to_scalar(v_log)
            lg = torch.nan_to_num(p_log[:, :A].float(), nan=0.0, posinf=float("inf"), neginf=float("-inf"))
            tree.backprop(sim + 1, cfg.discount, r.cpu().numpy(), v.cpu().numpy(), lg.cpu().numpy())
        dist, vals = tree.distributions(), tree.values()
        acts = np.zeros(N, np.int32)
        for i in range(N):
            a, ent, masked = ref_select_action(dist[i], legal[i], uni[i])
            acts[i] = a
            hist[i]["visits"].append(masked), hist[i]["value"].append(vals[i]), hist[i]["action"].append(a)
            assert abs(ent - float(actor.entropy[i])) < 1e-12
        assert (actor.action.cpu().numpy() == acts).all(), (step, actor.action.cpu().numpy(), acts)
        rew, done, score = oenv.step(acts)
        obs, legal = oenv.observe()
        for i in range(N):
            hist[i]["reward"].append(rew[i]), hist[i]["obs"].append(obs[i].copy()), hist[i]["legal"].append(legal[i].copy())
            windows[i] = windows[i][1:] + [obs[i].copy()]
        if done.any():
            oenv.reset(done)
            obs, legal = oenv.observe()
            for i in np.nonzero(done)[0]:
                finished.append((i, score[i], hist[i]))
                hist[i] = dict(obs=[obs[i].copy()], legal=[legal[i].copy()], action=[], reward=[], visits=[], value=[])
                windows[i] = [obs[i].copy() for _ in range(stack)]
    assert int(actor.illegal_steps) == 0
    rec = actor.drain()
    assert len(finished) > 0 and rec is not None and rec["meta"].shape[0] == len(finished)
    for g, (i, score, h) in enumerate(finished):  # flush order == env order within a step == our append order
        r = unpack_record(rec, g)
        assert (r["env_id"], r["score"], r["len"]) == (i, score, len(h["action"]))
        assert (r["action"] == h["action"]).all() and (r["reward"] == h["reward"]).all()
        assert (r["visits"] == np.array(h["visits"])).all()
        assert (r["value"].view(np.uint32) == np.array(h["value"], np.float32).view(np.uint32)).all()
        assert (r["legal"] == np.array(h["legal"])).all()
        gh = GameHistory.from_packed(r, None, cfg)
        assert len(gh) == r["len"] and gh.obs_history.shape == (r["len"] + stack, D)
        assert (gh.obs_history[stack - 1:] == np.array(h["obs"])).all()
        assert np.allclose(gh.child_visits.sum(1), 1.0) and gh.legal_actions.shape == (r["len"] + 1, A)


@pytest.mark.parametrize("game,N,sims,stack,steps,dtype,use_graph", [
    ("Hanabi-Full", 48, 50, 4, 30, torch.float16, True),     # the benched lock-step: fp16 fused engine, k_search, fused tail, hipGraph
    ("Hanabi-Small", 40, 12, 2, 45, torch.float16, True),    # (A = 11: byte-misaligned legal rows in the hand-over; several games per env)
    ("Hanabi-Full-5p", 24, 16, 4, 45, torch.float16, True),
    ("Hanabi-Full", 24, 12, 4, 20, torch.bfloat16, False)])  # the same kernels enqueued one by one
def test_benched_lock_step_matches_oracle_replay(game, N, sims, stack, steps, dtype, use_graph):
    """The exact lock-step bench.py times -- root inference (hipBLASLt + the fused MFMA tail), k_prepare, ONE persistent search
    kernel, k_move_tail_a / k_move_tail_b with the next move's draws, replayed as a hipGraph -- against an independent replay of
    every move through the ORACLE tree + ORACLE env + the numpy restatement of select_action, fed by the stand-alone recurrent
    inference (tests/oracle_replay.py).  The actor is only LOOKED at between lock-steps (noise / uniforms it drew for the coming
    move, its window and legal masks, the actions it chose); nothing of its state enters the oracle side except the draws."""
    from oracle.cport import OracleEnv, OracleTree
    from hanabizero_amd.game import GameHistory
    from hanabizero_amd.selfplay import unpack_record
    from tests.oracle_replay import oracle_search
    cfg, eng, actor = make(game, N, sims, stack, dtype, use_graph=use_graph, seed=3)
    assert eng.fused is not None and actor._tail_is_fused()
    A, D = cfg.action_space_size, cfg.obs_dim
    snaps = []

    def snap():
        torch.cuda.synchronize()
        snaps.append({k: getattr(actor, k).clone() for k in ("noise", "uniform", "stack_buf", "legal", "action", "entropy")})

    body = actor._step_body

    def looked_at(draw=True):  # (the two eager lock-steps inside the graph capture are moves like any other)
        body(draw)
        if not torch.cuda.is_current_stream_capturing():
            snap()

    actor._step_body = looked_at
    actor._draw()
    actor._drawn = True
    snap()
    while actor.total_moves < steps * N:
        actor.step()
        if actor.use_graph:
            snap()
    assert len(snaps) == steps + 1 and (not use_graph or actor._graph is not None)
    assert int(actor.illegal_steps) == 0
    # ---- the oracle plays the same moves
    oenv = OracleEnv(game, 3 + np.arange(N))
    oenv.reset()
    obs, legal = oenv.observe()
    windows = [[obs[i].copy() for _ in range(stack)] for i in range(N)]
    hist = [dict(obs=[obs[i].copy()], legal=[legal[i].copy()], action=[], reward=[], visits=[], value=[]) for i in range(N)]
    finished = []
    for step in range(steps):
        pre, post = snaps[step], snaps[step + 1]
        noise, uni = pre["noise"].cpu().numpy(), pre["uniform"].cpu().numpy()
        stack_in = np.stack([np.concatenate(w) for w in windows]).astype(np.float32)
        assert (pre["stack_buf"][:, :, :D].reshape(N, -1).float().cpu().numpy() == stack_in).all(), step
        assert (pre["legal"].cpu().numpy() == legal).all(), step
        win = np.zeros((N, stack, actor.Dp), np.float32)
        win[:, :, :D] = stack_in.reshape(N, stack, D)
        _, l0, h0 = eng.initial(torch.from_numpy(win.reshape(N, -1)).cuda(), padded=actor.Dp != D)
        tree = OracleTree(N, A, sims, seed=3, value_delta_max=cfg.value_delta_max)
        tree.prepare(cfg.root_exploration_fraction, noise, np.zeros(N, np.float32), l0.cpu().numpy(), legal)
        oracle_search(cfg, eng, tree, h0, sims)
        dist, vals = tree.distributions(), tree.values()
        acts = np.zeros(N, np.int32)
        for i in range(N):
            a, ent, masked = ref_select_action(dist[i], legal[i], uni[i])
            acts[i] = a
            hist[i]["visits"].append(masked), hist[i]["value"].append(vals[i]), hist[i]["action"].append(a)
            assert abs(ent - float(post["entropy"][i])) < 1e-12
        assert (post["action"].cpu().numpy() == acts).all(), (step, post["action"].cpu().numpy(), acts)
        rew, done, score = oenv.step(acts)
        obs, legal = oenv.observe()
        for i in range(N):
            hist[i]["reward"].append(rew[i]), hist[i]["obs"].append(obs[i].copy()), hist[i]["legal"].append(legal[i].copy())
            windows[i] = windows[i][1:] + [obs[i].copy()]
        if done.any():
            oenv.reset(done)
            obs, legal = oenv.observe()
            for i in np.nonzero(done)[0]:
                finished.append((i, score[i], hist[i]))
                hist[i] = dict(obs=[obs[i].copy()], legal=[legal[i].copy()], action=[], reward=[], visits=[], value=[])
                windows[i] = [obs[i].copy() for _ in range(stack)]
    rec = actor.drain()
    assert len(finished) > 0 and rec is not None and rec["meta"].shape[0] == len(finished)
    for g, (i, score, h) in enumerate(finished):
        r = unpack_record(rec, g)
        assert (r["env_id"], r["score"], r["len"]) == (i, score, len(h["action"]))
        assert (r["action"] == h["action"]).all() and (r["reward"] == h["reward"]).all()
        assert (r["visits"] == np.array(h["visits"])).all()
        assert (r["value"].view(np.uint32) == np.array(h["value"], np.float32).view(np.uint32)).all()
        assert (r["legal"] == np.array(h["legal"])).all()
        gh = GameHistory.from_packed(r, None, cfg)
        assert (gh.obs_history[stack - 1:] == np.array(h["obs"])).all()


def test_graph_replay_equals_eager():
    """The hipGraph-captured lock-step produces exactly what the eager stream of the same kernels produces."""
    outs = []
    for use_graph in (False, True):
        cfg, eng, actor = make("Hanabi-Small", 64, 10, 2, torch.bfloat16, use_graph, seed=11)
        for _ in range(30 if use_graph else 32):
            actor.step()
        torch.cuda.synchronize()
        rec = actor.drain()
        outs.append((actor.action.cpu().numpy().copy(), actor.env.probe().cpu().numpy().copy(), rec))
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()
    assert outs[0][2]["meta"].shape == outs[1][2]["meta"].shape
    for k in outs[0][2]:
        assert (outs[0][2][k] == outs[1][2][k]).all(), k


@pytest.mark.parametrize("game,dtype", [("Hanabi-Small", torch.bfloat16), ("Hanabi-Full", torch.float16)])
def test_engine_load_reaches_a_graph_captured_actor(game, dtype):
    """A weight update (net.set_weights(w); engine.load(net): selfplay_worker.py:177-184) must reach an actor whose lock-step
    is a captured hipGraph -- the graph has the engine's tensor addresses baked in, so load() overwrites them in place.
    A graph actor and an eager actor (which reads the engine's tensors at call time: the ground truth) play 5 moves,
    both engines load new weights, they play 4 more: everything they record must agree, and must differ from a graph
    actor that kept the old weights."""
    import copy
    runs = []
    for use_graph, reload in ((True, True), (False, True), (True, False)):
        cfg, eng, actor = make(game, 64, 12, 2, dtype, use_graph, seed=21)
        ptrs = sorted(t.data_ptr() for t in eng._dev.values())
        for _ in range(3 if use_graph else 5):  # (capturing plays two warm-up moves of its own: 5 moves either way)
            actor.step()
        if reload:
            net2 = copy.deepcopy(eng._net)
            g = torch.Generator().manual_seed(5)
            with torch.no_grad():
                for p in net2.parameters():
                    p.add_(0.02 * torch.randn(p.shape, generator=g))
            v_before = eng.version
            eng.load(net2)
            assert eng.version == v_before + 1
            assert sorted(t.data_ptr() for t in eng._dev.values()) == ptrs, "load() moved a device tensor"
        for _ in range(4):
            actor.step()
        torch.cuda.synchronize()
        assert int(actor.illegal_steps) == 0
        runs.append((actor.counts.clone(), actor.values.clone(), actor.action.clone(), actor.pool.clone(),
                     actor.traj["action"].clone(), actor.traj_len.clone(), actor.env.probe().clone()))
    bits = lambda t: t.view(torch.int16) if t.dtype in (torch.float16, torch.bfloat16) else (t.view(torch.int32) if t.dtype == torch.float32 else t)
    for a, b in zip(runs[0], runs[1]):  # (bit patterns: fp16 hidden states of the perturbed net may hold inf / NaN)
        assert torch.equal(bits(a), bits(b)), "the graph actor did not see the weights the eager actor saw"
    assert not torch.equal(runs[0][0], runs[2][0]) and not torch.equal(bits(runs[0][3]), bits(runs[2][3])), "the new weights changed nothing"


def test_root_noise_stream_is_dirichlet_and_sharding_independent():
    """hz_actor_draw: Dirichlet(alpha) rows (marginals Beta(alpha, (A-1) alpha), KS test), U[0,1) uniforms, and the
    draws of env i at move k depend only on (seed, global env id, k)."""
    from scipy import stats
    from hanabizero_amd._lib import check, lib
    N, A, alpha = 8192, 20, 0.3
    s = torch.cuda.current_stream().cuda_stream

    def draw(base, n, moves):
        mc = torch.zeros(n, dtype=torch.int64, device="cuda")
        noise = torch.zeros(n, A, dtype=torch.float32, device="cuda")
        uni = torch.zeros(n, dtype=torch.float64, device="cuda")
        out = []
        for _ in range(moves):
            check(lib.hz_actor_draw(77, base, mc.data_ptr(), n, A, alpha, noise.data_ptr(), uni.data_ptr(), s), "draw")
            out.append((noise.cpu().numpy().copy(), uni.cpu().numpy().copy()))
        assert (mc == moves).all()
        return out

    full = draw(0, N, 3)
    noise, uni = full[0]
    assert np.isfinite(noise).all() and (noise >= 0).all() and np.abs(noise.sum(1) - 1).max() < 1e-5
    assert (uni >= 0).all() and (uni < 1).all()
    assert stats.kstest(uni, "uniform").pvalue > 1e-3
    for a in (0, 7, A - 1):
        assert stats.kstest(noise[:, a].astype(np.float64), stats.beta(alpha, (A - 1) * alpha).cdf).pvalue > 1e-3, a
    assert abs(noise.mean() - 1.0 / A) < 1e-3
    var = alpha * (A * alpha - alpha) / ((A * alpha) ** 2 * (A * alpha + 1))
    assert abs(noise.var(0).mean() - var) / var < 0.05
    assert abs(np.corrcoef(noise[:, 0], noise[:, 1])[0, 1] + 1.0 / (A - 1)) < 0.05  # Dirichlet: -alpha/(alpha0 - alpha)
    assert not (full[0][0] == full[1][0]).all()                                      # next move, new draws
    part = draw(4096, 100, 3)                                                         # another sharding of the same envs
    for k in range(3):
        assert (part[k][0] == full[k][0][4096:4196]).all() and (part[k][1] == full[k][1][4096:4196]).all()


def test_select_action_kernel_edge_cases():
    from hanabizero_amd._lib import check, lib
    N, A = 5, 11
    counts = torch.tensor([[0] * 10 + [9], [3] * 11, [0] * 11, [5, 4] + [0] * 9, [1] * 11], dtype=torch.int32, device="cuda")
    legal = torch.ones(N, A, dtype=torch.uint8, device="cuda")
    legal[3, 0] = 0
    u = torch.tensor([0.999, 0.0, 0.5, 0.0, 1.0 - 1e-16], dtype=torch.float64, device="cuda")
    act = torch.zeros(N, dtype=torch.int32, device="cuda")
    ent = torch.zeros(N, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    check(lib.hz_select_action(N, A, counts.data_ptr(), legal.data_ptr(), u.data_ptr(), 1.0, 0, act.data_ptr(), ent.data_ptr(), s), "x")
    assert act.tolist() == [10, 0, -1, 1, 10]
    assert counts[3].tolist() == [0, 4] + [0] * 9  # illegal count zeroed in place
    assert abs(float(ent[1]) - np.log2(11)) < 1e-12 and float(ent[0]) == 0.0
    check(lib.hz_select_action(N, A, counts.data_ptr(), legal.data_ptr(), u.data_ptr(), 1.0, 1, act.data_ptr(), None, s), "x")
    assert act.tolist() == [10, 0, -1, 1, 0]


@pytest.mark.parametrize("game,N,sims,peaked,dtype", [
    ("Hanabi-Small", 100, 12, False, torch.bfloat16), ("Hanabi-Full", 50, 50, False, torch.bfloat16),
    ("Hanabi-Full", 1000, 20, False, torch.bfloat16), ("Hanabi-Full-5p", 70, 30, False, torch.bfloat16),
    ("Hanabi-Full", 4170, 8, False, torch.bfloat16), ("Hanabi-Full", 45, 50, True, torch.bfloat16),
    ("Hanabi-Small", 100, 12, False, torch.float16), ("Hanabi-Full", 1000, 20, False, torch.float16),
    ("Hanabi-Full-5p", 70, 30, False, torch.float16), ("Hanabi-Full", 45, 50, True, torch.float16),
    ("Hanabi-Full", 700, 50, "sharp", torch.bfloat16), ("Hanabi-Full", 700, 50, "sharp", torch.float16),
    ("Hanabi-Full-5p", 300, 50, "sharp", torch.float16)])  # (A = 48: no predicted lines there, long paths all the same)
def test_persistent_search_kernel_equals_launch_per_phase(game, N, sims, peaked, dtype):
    """hz_search_run (all simulations in one persistent kernel, a workgroup per 16 trees) against the launch-per-phase
    search (hz_tree_traverse -> hz_mlp_recurrent -> hz_tree_backprop_traverse ...): bit-identical trees, hidden-state
    pools and leaf outputs; the launch-per-phase path itself is pinned to the oracle by the tests above."""
    from hanabizero_amd import cytree
    from hanabizero_amd.mcts import MCTS
    from hanabizero_amd._lib import poll_giveups
    giveups_before = poll_giveups()
    cfg, eng, actor = make(game, N, sims, 2, dtype, use_graph=False, peaked=peaked)
    assert eng.fused is not None and eng.fused.header.dtype == {torch.bfloat16: 1, torch.float16: 2}[dtype]
    A = cfg.action_space_size
    g = torch.Generator(device="cuda").manual_seed(N)
    value0, logits0, hidden0 = actor.root_inference()
    noise = torch.rand(N, A, device="cuda", generator=g)
    noise = noise / noise.sum(1, keepdim=True)
    res = []
    # launch per phase; the persistent kernel with 16 trees per workgroup: one per wavefront, or two side by side in the
    # halves of 8 wavefronts (-16; A <= 32); with 32: side by side or one after the other (-32; what A = 48 gets either way);
    # the library's own choice (32 once the trees outnumber 16 per compute unit: the last case)
    # ... and the 16- / 32-tree kernels once more without the descent along predicted lines (hz_search_set_predicted_lines: "plain")
    for persistent in (False, 16, -16, 32, -32, "auto", "plain16", "plain32"):
        roots = cytree.Roots(N, A, sims, tie_seed=5, tree_id_base=17)
        if str(persistent).startswith("plain"):
            roots.set_predicted_lines(False)
            persistent = int(persistent[5:])
        roots.prepare(0.0 if peaked is True else cfg.root_exploration_fraction, noise, torch.zeros(N, device="cuda"), logits0, actor.legal)
        pool = torch.zeros(sims, N, eng.H, dtype=eng.dtype, device="cuda")
        MCTS(cfg, persistent=bool(persistent), rows_per_workgroup=0 if persistent in (False, "auto") else int(persistent)
             ).run_multi(roots, eng, hidden0, pool=pool)
        torch.cuda.synchronize()
        res.append((roots.distributions_tensor(), roots.values_tensor(), roots.trajectories_tensor(),
                    roots.minmax_tensors(), roots.path_len_tensor(), pool))
    a = res[0]
    for b in res[1:]:
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[4], b[4])
        assert torch.equal(a[3][0], b[3][0]) and torch.equal(a[3][1], b[3][1])
        assert torch.equal(a[5].view(torch.int16), b[5].view(torch.int16))  # (bit patterns: a 49-deep fp16 chain of random nets overflows to inf / NaN)
    assert int(a[0].sum()) == N * (sims - 1)
    assert poll_giveups() == giveups_before, "a wave gave up waiting for an arrival counter (include/hz_mlp.h)"
    if peaked == "sharp":  # (the 16-tree kernel walks long repeated paths sixteen levels at a time: hz_tree_replay_dev.h)
        assert int(a[4].max()) > (12 if game == "Hanabi-Full" else 8) and float(a[4].float().mean()) > 4, (int(a[4].max()), float(a[4].float().mean()))
    if peaked is True and dtype == torch.bfloat16:  # paths longer than the 32 lanes a tree has in the side-by-side kernel: its backup
        assert int(a[4].max()) > 34, int(a[4].max())  # runs in two chunks (in fp16 the 49-deep chain of random nets turns NaN first)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_config3_at_full_size_8192_envs_50_simulations(dtype):
    """BASELINE.json configs[2] at its own size through the product path: 8192 Hanabi-Full envs, 50 simulations per move, the
    self-play actor under its hipGraph with the search kernel shape the library chooses there (32 trees per workgroup, two
    side by side per wave: k_search_half).  Size-independent properties: every tree's visit counts add up to 49, no illegal
    env step, no wait on an arrival counter given up; and on the same prepared roots the forced 16-trees-per-workgroup shape
    -- the one the other tests pin to the oracle through the launch-per-phase search -- gives the same bits."""
    from hanabizero_amd import cytree
    from hanabizero_amd.mcts import MCTS
    from hanabizero_amd._lib import poll_giveups
    N, sims = 8192, 50
    giveups_before = poll_giveups()
    cfg, eng, actor = make("Hanabi-Full", N, sims, 4, dtype, use_graph=True, seed=8)
    for move in range(3):
        actor.step()
        torch.cuda.synchronize()
        dist = actor.roots.distributions_tensor()
        assert dist.shape == (N, cfg.action_space_size) and bool((dist.sum(1) == sims - 1).all()), move
        masked = actor.counts
        assert bool(((masked == dist) | ((masked == 0) & (dist >= 0))).all()) and bool((masked.sum(1) > 0).all())
    # (the first step() also ran the two eager lock-steps that precede the graph capture)
    assert int(actor.illegal_steps) == 0 and int(actor.traj_len.max()) == actor.total_moves // N == 5 and int(actor.traj_len.min()) >= 0
    # the same roots, searched by the shape of the library's choice and by the forced 16-row shape
    A = cfg.action_space_size
    _, logits0, hidden0 = actor.root_inference()
    noise = actor.noise.clone()
    res = []
    for rows in (0, 16):
        roots = cytree.Roots(N, A, sims, tie_seed=9, tree_id_base=3)
        roots.prepare(cfg.root_exploration_fraction, noise, torch.zeros(N, device="cuda"), logits0, actor.legal)
        pool = torch.zeros(sims, N, eng.H, dtype=eng.dtype, device="cuda")
        MCTS(cfg, persistent=True, rows_per_workgroup=rows).run_multi(roots, eng, hidden0, pool=pool)
        torch.cuda.synchronize()
        res.append((roots.distributions_tensor(), roots.values_tensor(), roots.path_len_tensor(), pool))
        del roots
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    assert torch.equal(res[0][3].view(torch.int16), res[1][3].view(torch.int16))
    assert int(res[0][0].sum()) == N * (sims - 1)
    assert poll_giveups() == giveups_before, "a wave gave up waiting for an arrival counter (include/hz_mlp.h)"


def test_persistent_search_limits_fall_back_to_launch_per_phase():
    """hz_search_run refuses 64 or more simulations (include/hz_search.h); MCTS.run_multi then runs the launch-per-phase
    search, whose results are what they would have been."""
    from hanabizero_amd import cytree
    from hanabizero_amd._lib import HzError
    from hanabizero_amd.mcts import MCTS
    sims, N = 70, 48
    cfg, eng, actor = make("Hanabi-Small", N, sims, 2, torch.bfloat16, use_graph=False)
    A = cfg.action_space_size
    value0, logits0, hidden0 = actor.root_inference()
    noise = torch.full((N, A), 1.0 / A, device="cuda")
    res = []
    for persistent in (True, False):
        roots = cytree.Roots(N, A, sims, tie_seed=3)
        roots.prepare(cfg.root_exploration_fraction, noise, torch.zeros(N, device="cuda"), logits0, actor.legal)
        MCTS(cfg, persistent=persistent).run_multi(roots, eng, hidden0)
        res.append(roots.distributions_tensor())
    assert torch.equal(res[0], res[1]) and int(res[0].sum()) == N * (sims - 1)
    roots = cytree.Roots(N, A, sims, tie_seed=3)
    roots.prepare(cfg.root_exploration_fraction, noise, torch.zeros(N, device="cuda"), logits0, actor.legal)
    roots.set_params(cfg.pb_c_base, cfg.pb_c_init, cfg.discount, cfg.value_delta_max)
    pool = torch.zeros(sims, N, eng.H, dtype=eng.dtype, device="cuda")
    pool[0].copy_(hidden0)
    rew, val, pol = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, A, device="cuda")
    with pytest.raises(HzError, match="64 simulations"):
        roots.search_tensors(eng.fused_shape(16, 2), pool, sims - 1, rew, val, pol, 0)


def test_packed_drain_equals_drain():
    """drain_packed (one ragged byte buffer that stays on the device, hz_actor_pack) carries exactly the games drain()
    returns, and the host-side packer (pack_records) produces the same bytes section by section."""
    from hanabizero_amd.selfplay import pack_records, packed_layout, unpack_packed, unpack_record
    recs = []
    for packed in (False, True):
        cfg, eng, actor = make("Hanabi-Small", 64, 10, 2, torch.bfloat16, True, seed=33)
        for rnd in range(2):  # two drains: the second starts in the middle of the outbox ring
            for _ in range(25):
                actor.step()
            torch.cuda.synchronize()
            if packed:
                buf, n, moves = actor.drain_packed()
                assert buf.is_cuda and buf.numel() == packed_layout(n, moves, actor.A, actor.W)[1]
                recs.append(unpack_packed(buf.cpu().numpy(), n, moves, actor.A, actor.W))
                assert actor.drain_packed() is None
            else:
                recs.append(actor.drain())
    for padded, ragged in ((recs[0], recs[2]), (recs[1], recs[3])):
        n = padded["meta"].shape[0]
        assert n > 10 and ragged["meta"].shape[0] == n and (padded["meta"] == ragged["meta"]).all()
        assert len(set(padded["meta"][:, 0].tolist())) > 1  # games of different lengths
        for i in range(n):
            a, b = unpack_record(padded, i), unpack_record(ragged, i)
            for k in a:
                assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (i, k)
        hbuf, hn, hmoves = pack_records(padded, actor.A, actor.W)
        again = unpack_packed(hbuf, hn, hmoves, actor.A, actor.W)
        for k in again:
            assert again[k].dtype == ragged[k].dtype and np.array_equal(again[k], ragged[k]), k


@pytest.mark.parametrize("game,N,stack,dtype,moves,max_moves", [
    ("Hanabi-Small", 96, 2, torch.bfloat16, 28, None), ("Hanabi-Full", 70, 4, torch.float16, 28, None),
    ("Hanabi-Full-5p", 37, 4, torch.float32, 70, None),
    # (enough envs that the slot prefix of a late workgroup takes several trips)
    ("Hanabi-Small", 2501, 1, torch.float16, 9, None),
    # trajectory rows whose lengths are not multiples of 4 bytes (A = 11: legal rows of 6 * 11 = 66 B at byte-misaligned addresses,
    # action / reward rows of 5 or 13 B) in games that DO reach the last row (longer ones clamp to it, the same way in all
    # three forms): the hand-over must move every byte of every row (r03's copy_rows_block dropped the last n % 4)
    # (Hanabi-Small has one life: random-init play ends most games within five moves; Hanabi-Full's action rows are 13 B here)
    ("Hanabi-Small", 96, 2, torch.bfloat16, 20, 5), ("Hanabi-Full", 70, 4, torch.float16, 40, 13)])
def test_fused_launches_equal_their_separate_calls(game, N, stack, dtype, moves, max_moves):
    """The lock-step's tail three ways: (tail) the two launches of include/hz_movetail.h, one wave per env from the root read-out
    to the next move's window; (fused) one launch per phase with the fusions of r01 (hz_actor_begin_move_draw; hz_env_reset_rows
    carrying the flush); (separate) every entry point on its own (hz_actor_draw + hz_actor_begin_move; hz_actor_flush +
    hz_env_reset).  The same finished games, bit for bit, and the same live state -- histories under construction, windows,
    legal masks, env states and generators, the next move's noise and uniforms, the outbox counters."""
    from hanabizero_amd._lib import check, lib
    import ctypes as C
    recs, states = [], []
    for form in ("tail", "fused", "separate"):
        cfg, eng, actor = make(game, N, 10, stack, dtype, use_graph=False, seed=21, fused_tail=form == "tail", max_moves=max_moves)
        if form == "separate":
            def reset_then(mask, rows=None, _env=actor.env, _actor=actor):  # flush and reset as two launches
                check(lib.hz_actor_flush(C.byref(_actor.bufs), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "flush")
                type(_env).reset(_env, mask)
            actor.env.reset = reset_then
        for _ in range(moves):
            if form != "separate":
                actor._step_body(draw=True)
            else:
                actor._draw()
                actor._step_body(draw=False)
        if form == "separate":
            actor._draw()  # (the other two forms have drawn for the next move already)
        torch.cuda.synchronize()
        live = dict(stack=actor.stack_buf, legal=actor.legal, traj_len=actor.traj_len, ent_sum=actor.ent_sum, probe=actor.env.probe(),
                    noise=actor.noise, uniform=actor.uniform, move_count=actor.move_count, out_count=actor.out_count,
                    counts=actor.counts, values=actor.values, action=actor.action, slot=actor.slot,
                    num_finished=actor.num_finished, illegal=actor.illegal_steps, **{"traj_" + k: v for k, v in actor.traj.items()})
        n_out = min(int(actor.out_count[0]), actor.cap)  # the outbox slots in use, whole rows: every byte the hand-over moved
        live.update({"out_" + k: v[:n_out] for k, v in actor.out.items()})
        live["out_meta"] = actor.out_meta[:n_out]
        states.append({k: v.clone() for k, v in live.items()})
        recs.append(actor.drain())
    assert recs[0]["meta"].shape[0] > (20 if game == "Hanabi-Small" else 3)
    if max_moves is not None:  # (the case is about the last row: games did reach it, and its last bytes are not all zero)
        assert int(recs[0]["meta"][:, 0].max()) >= max_moves
        assert int(states[0]["out_legal"][:, max_moves, -3:].sum()) > 0 and int(states[0]["out_action"][:, max_moves - 1].abs().sum()) > 0
    for other in (1, 2):
        for k in recs[0]:
            assert np.array_equal(recs[0][k], recs[other][k]), (other, k)
        for k in states[0]:
            assert torch.equal(states[0][k], states[other][k]), (other, k)
    assert int(states[0]["illegal"]) == 0


def test_temperature_schedule_reaches_a_graph_captured_actor():
    """config.visit_softmax_temperature_fn (selfplay_worker.py:172-174) changes with the learner's step counter when
    change_temperature is on.  The fused tail reads the temperature from device memory when it runs, so an actor whose lock-step
    was captured at temperature 1 samples at 0.5 after set_trained_steps -- exactly like an eager actor that enqueues the
    launch-per-phase tail with the new value."""
    recs = []
    for use_graph, fused_tail in ((True, True), (False, False)):
        cfg, eng, actor = make("Hanabi-Small", 80, 10, 2, torch.float16, use_graph=use_graph, seed=13, fused_tail=fused_tail)
        cfg.visit_softmax_temperature_fn = lambda num_moves, trained_steps: 1.0 if trained_steps < 100 else 0.5
        actor.set_trained_steps(0)
        n_first = 9 if use_graph else 11  # (the graph-captured actor's first step() also runs the two eager warm-up lock-steps)
        for _ in range(n_first):
            actor.step()
        actor.set_trained_steps(200)
        for _ in range(14):
            actor.step()
        torch.cuda.synchronize()
        assert actor.total_moves == 25 * 80 and float(actor.temperature) == 0.5
        recs.append((actor.drain(), actor.action.clone(), actor.entropy.clone()))
    assert recs[0][0]["meta"].shape[0] > 10
    for k in recs[0][0]:
        assert np.array_equal(recs[0][0][k], recs[1][0][k]), k
    assert torch.equal(recs[0][1], recs[1][1]) and torch.equal(recs[0][2], recs[1][2])


def test_new_entry_points_report_bad_arguments():
    """hz_actor_pack / hz_actor_packed_bytes / hz_env_reset_rows / hz_actor_begin_move_draw / hz_search_run:
    malformed calls come back as error codes with a message (the reference aborts or corrupts memory), nothing is launched."""
    import ctypes as C
    from hanabizero_amd._lib import HzError, RowsJob, check, lib
    cfg, eng, actor = make("Hanabi-Small", 32, 10, 2, torch.bfloat16, use_graph=False)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    b = C.byref(actor.bufs)
    out = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    starts = torch.zeros(64, dtype=torch.int32, device="cuda")
    assert lib.hz_actor_packed_bytes(-1, 5, actor.A, actor.W, None) == -1
    assert lib.hz_actor_packed_bytes(3, 7, actor.A, actor.W, None) % 16 == 0
    for args in ((0, 0, 5), (0, 5, 2), (0, 2, 10 ** 6), (-1, 2, 5)):  # n = 0; moves < n; moves > n * T; first < 0
        with pytest.raises(HzError):
            check(lib.hz_actor_pack(b, args[0], args[1], args[2], starts.data_ptr(), out.data_ptr(), out.numel(), st), "pack")
    with pytest.raises(HzError):  # buffer too small
        check(lib.hz_actor_pack(b, 0, 2, 9, starts.data_ptr(), out.data_ptr(), 64, st), "pack")
    job = RowsJob()
    check(lib.hz_actor_flush_job(b, C.byref(job)), "job")
    assert job.num_arrays == 7 and job.max_rows == 32 and job.row_bytes[6] == 16
    job.num_arrays = 9
    with pytest.raises(HzError):
        check(lib.hz_env_reset_rows(actor.env._h, None, C.byref(job), st), "reset_rows")
    job.num_arrays = 7
    job.row_bytes[2] = 0
    with pytest.raises(HzError):
        check(lib.hz_env_reset_rows(actor.env._h, None, C.byref(job), st), "reset_rows")
    es = actor.newest.element_size()
    with pytest.raises(HzError):  # alpha <= 0
        check(lib.hz_actor_begin_move_draw(b, actor.env.done.data_ptr(), actor.tmp_packed.data_ptr(), actor.legal.data_ptr(),
                                           actor.newest.data_ptr(), actor.newest.stride(0) * es, actor.stack_buf.data_ptr(),
                                           actor.stack_buf.stride(0) * es, actor.stack, actor.Dp * es, 1,
                                           actor.move_count.data_ptr(), 0.0, actor.noise.data_ptr(), actor.uniform.data_ptr(), st),
              "begin_move_draw")
    from hanabizero_amd import cytree
    roots = cytree.Roots(32, actor.A, actor.S)
    roots.prepare(0.25, actor.noise, actor.zeros_n, torch.zeros(32, actor.A, device="cuda"), actor.legal)
    roots.set_params(cfg.pb_c_base, cfg.pb_c_init, cfg.discount, cfg.value_delta_max)
    rew, pol = torch.zeros(32, device="cuda"), torch.zeros(32, actor.A, device="cuda")
    with pytest.raises(HzError):  # rows per workgroup: 0, 16, 32 or -32
        roots.search_tensors(eng.fused_shape(16, 2), actor.pool, actor.S - 1, rew, rew, pol, rows_per_workgroup=48)
    torch.cuda.synchronize()
    for _ in range(3):  # and the actor is still in working order
        actor.step()
    torch.cuda.synchronize()
    assert int(actor.illegal_steps) == 0


def test_actor_follows_path_lengths_to_the_predicted_line_kernels():
    """predicted_lines="auto": an actor starts with the plain search kernels and changes to the predicted-line ones at the first
    drain that sees long last paths (a sharp policy), captures its lock-step again, and stays there; with random-init nets it
    stays plain.  The games do not depend on the kernels: after the same number of moves the same env states, windows and last
    actions as an actor that used the other kernels throughout."""
    for peaked, expect in (("sharp", True), (False, False)):
        out = []
        for mode in ("auto", not expect):
            cfg, eng, actor = make("Hanabi-Full", 256, 50, 4, torch.float16, use_graph=True, peaked=peaked)
            actor.predicted_lines = mode
            actor._lines_on = mode is True
            actor.roots.set_predicted_lines(actor._lines_on)
            calls = 0
            while actor.total_moves < 26 * actor.N:  # (a capture plays two moves of its own: count moves, not calls)
                actor.step()
                calls += 1
                if calls % 5 == 0:
                    actor.drain_packed()
            torch.cuda.synchronize()
            assert actor.total_moves == 26 * actor.N
            if mode == "auto":
                assert actor._lines_on == expect, float(actor.roots.path_len_tensor().float().mean())
            out.append((actor.action.clone(), actor.env.snapshot(), actor.stack_buf.clone()))
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][2], out[1][2])
        for x, y in zip(out[0][1], out[1][1]):
            assert torch.equal(x, y)
