"""The learner step (hanabizero_amd/learner.py): targets, gradient shaping and the optimiser step against
independent restatements in fp32 on the CPU; the bf16-autocast step on the GPU.  (The step itself and make_batch are pinned
to the reference's own update_weights / batch workers by tests/test_reference_callers.py; these tests cover the pieces.)"""
import numpy as np
import pytest
import torch


def _cfg(game="Hanabi-Small", stack=2):
    from hanabizero_amd.config import make_config
    return make_config(game, simulations=10, stack=stack, batch_size=8)


def _batch(cfg, B, seed):
    rng = np.random.RandomState(seed)
    U, A, D, stack = cfg.num_unroll_steps, cfg.action_space_size, cfg.obs_dim, cfg.stacked_observations
    obs = (rng.rand(B, stack + U, D) < 0.2).astype(np.float32)
    actions = rng.randint(0, A, (B, U))
    mask = np.ones((B, U), np.float32)
    pol = rng.rand(B, U + 1, A).astype(np.float32)
    pol /= pol.sum(-1, keepdims=True)
    inputs = (obs, actions, mask, np.arange(B), rng.rand(B).astype(np.float32) + 0.5, np.zeros(B))
    targets = (rng.randint(-1, 2, (B, U)).astype(np.float32), (rng.rand(B, U + 1) * 20 - 5).astype(np.float32), pol)
    return inputs, targets


def test_scalar_transform_and_two_hot_targets():
    from hanabizero_amd.learner import phi, scalar_transform
    from hanabizero_amd.model import inverse_scalar_transform
    x = torch.tensor([[-7.3, -1.0, 0.0, 0.4, 2.0, 55.5, 2999.0]])
    h = scalar_transform(x)
    want = torch.sign(x) * (torch.sqrt(x.abs() + 1) - 1) + 0.001 * x  # Appendix F of the MuZero paper, eps = 0.001
    assert torch.allclose(h, want, atol=1e-6) and float(h[0, 2]) == 0.0
    t = phi(h, -25, 25, 51)
    assert t.shape == (1, 7, 51) and torch.allclose(t.sum(-1), torch.ones(1, 7))
    hc = h.clamp(-25, 25)
    assert torch.allclose((t * torch.arange(-25, 26.0)).sum(-1), hc, atol=1e-5)      # expectation = clamped h(x)
    assert (t > 0).sum(-1).max() <= 2 and float(t[0, 2, 25]) == 1.0                   # two-hot; integer -> one-hot
    assert float(t[0, 3, 25 + 0]) == pytest.approx(1 - float(h[0, 3]), abs=1e-6)      # 0 < h(0.4) < 1: mass on 0 and 1
    # the categorical heads' read-out inverts it: softmax(log two-hot) . support -> h^-1 -> x  (core/config.py:204-232)
    back = inverse_scalar_transform(torch.log(t[0] + 1e-30), -25, 25).reshape(-1)
    assert torch.allclose(back[:6], x[0, :6], rtol=2e-3, atol=2e-3)


def test_losses_and_gradient_shaping_match_an_independent_restatement():
    """compute_losses against a re-derivation that scales gradients with x * s + x.detach() * (1 - s) instead of hooks."""
    from hanabizero_amd.learner import compute_losses, phi, scalar_transform
    cfg = _cfg()
    torch.manual_seed(0)
    net = cfg.get_uniform_network()
    for p in net.parameters():  # the reference zero-initialises the heads' last layers: give every layer a gradient path
        if float(p.detach().abs().sum()) == 0.0:
            torch.nn.init.normal_(p, std=0.05)
    net.train()
    (obs, act, _, _, w, _), (tr, tv, tp) = _batch(cfg, 8, 1)
    obs_t, act_t = torch.from_numpy(obs[:, :cfg.stacked_observations]), torch.from_numpy(act)
    tr_t, tv_t, tp_t, w_t = map(torch.from_numpy, (tr, tv, tp, w))
    loss, parts = compute_losses(net, cfg, obs_t, act_t, tr_t, tv_t, tp_t, w_t, amp=None)
    gscale = 1.0 / cfg.num_unroll_steps
    loss.register_hook(lambda g: g * gscale)
    net.zero_grad()
    loss.backward()
    got = [p.grad.clone() for p in net.parameters()]

    def sg(x, s):
        return x * s + x.detach() * (1 - s)
    B, U = 8, cfg.num_unroll_steps
    ce = lambda logits, target: -(torch.log_softmax(logits, 1) * target).sum(1)
    rphi = phi(scalar_transform(tr_t), -25, 25, 51)
    vphi = phi(scalar_transform(tv_t), -25, 25, 51)
    state = net.representation(obs_t.reshape(B, -1))
    logits, value = net.prediction(state)
    vl, pl, rl = ce(value, vphi[:, 0]), ce(logits, tp_t[:, 0]), torch.zeros(B)
    for k in range(U):
        # train.py:169 hooks the tensor recurrent_inference returns, and the step's own reward / value / policy heads
        # read that same tensor (core/model.py:74-84): everything that flows back into it is halved, this step's heads
        # included -- so the scaling sits between the dynamics trunk and the heads
        one_hot = torch.zeros(B, cfg.action_space_size).scatter_(1, act_t[:, k:k + 1], 1.0)
        state = sg(net._dynamics_state(torch.cat((state, one_hot), 1)), 0.5)
        reward = net._dynamics_reward(state)
        logits, value = net.prediction(state)
        pl, vl, rl = pl + ce(logits, tp_t[:, k + 1]), vl + ce(value, vphi[:, k + 1]), rl + ce(reward, rphi[:, k])
    ref = (w_t * (pl + cfg.value_loss_coeff * vl + rl)).mean()
    assert torch.allclose(loss, ref, rtol=1e-5) and torch.allclose(parts["value_loss"], vl, rtol=1e-5)
    net.zero_grad()
    sg(ref, gscale).backward()
    for g, p in zip(got, net.parameters()):
        assert torch.allclose(g, p.grad, rtol=1e-4, atol=1e-7)
    assert any(float(g.abs().sum()) > 0 for g in got)


def test_update_weights_descends_and_schedule():
    from hanabizero_amd.learner import adjust_lr, make_optimizer, update_weights
    cfg = _cfg()
    torch.manual_seed(1)
    net = cfg.get_uniform_network()
    opt = make_optimizer(net, cfg)
    assert opt.defaults["momentum"] == 0.9 and opt.defaults["weight_decay"] == 1e-4
    cfg.lr_init = 0.05
    assert adjust_lr(cfg, opt, 0) == 0.0 and adjust_lr(cfg, opt, cfg.lr_warm_step // 2) == pytest.approx(0.025)
    assert adjust_lr(cfg, opt, cfg.lr_warm_step + 5) == pytest.approx(0.05)
    batch = _batch(cfg, 8, 2)
    before = [p.detach().clone() for p in net.parameters()]
    losses = []
    for _ in range(12):
        loss_data, prio = update_weights(net, batch, opt, cfg, amp=None)
        losses.append(loss_data[1])
    assert np.isfinite(losses).all() and losses[-1] < 0.8 * losses[0]
    assert prio.shape == (8,) and (prio > 0).all()
    assert any(not torch.equal(a, b) for a, b in zip(before, net.parameters()))
    total = torch.sqrt(sum((p.grad ** 2).sum() for p in net.parameters()))
    assert float(total) <= cfg.max_grad_norm * 1.001  # clip_grad_norm_ leaves the clipped gradients behind


def test_make_batch_targets_follow_the_reference_recipe():
    from hanabizero_amd.learner import make_batch
    cfg = _cfg()
    U, td, A, D, stack, g = cfg.num_unroll_steps, cfg.td_steps, cfg.action_space_size, cfg.obs_dim, cfg.stacked_observations, cfg.discount
    rng = np.random.RandomState(3)

    class G:  # the GameHistory surface make_batch touches
        def __init__(self, T):
            self.actions = rng.randint(0, A, T)
            self.rewards = rng.randint(0, 2, T).astype(np.float64)
            self.child_visits = rng.dirichlet(np.ones(A), T)
            self.obs_history = (rng.rand(T + stack, D) < 0.3).astype(np.float32)

        def __len__(self):
            return len(self.actions)

        def obs(self, i, extra_len=0, padding=False):
            fr = self.obs_history[i:i + stack + extra_len]
            if padding and len(fr) < stack + extra_len:
                fr = np.concatenate((fr, np.repeat(fr[-1:], stack + extra_len - len(fr), 0)))
            return fr
    games, pos = [G(12), G(7)], [2, 5]
    value_fn = lambda o: o.sum(1) * 0.01
    (obs, act, mask, idx, w, _), (tr, tv, tp) = make_batch(games, pos, cfg, value_fn, rng=np.random.RandomState(0))
    assert obs.shape == (2, stack + U, D) and act.shape == (2, U) and tr.shape == (2, U + 1) and tp.shape == (2, U + 1, A)
    assert (act[0] == games[0].actions[2:2 + U]).all() and mask[1].tolist() == [1, 1, 0, 0, 0]
    assert (obs[1, -1] == games[1].obs_history[-1]).all()                       # padded with the last frame
    for b, (G_, p) in enumerate(zip(games, pos)):
        for j in range(U + 1):
            cur = p + j
            if cur >= len(G_):
                assert tv[b, j] == 0 and tr[b, j] == 0 and (tp[b, j] == 0).all()
                continue
            boot = cur + td
            v = 0.0
            if boot < len(G_):
                v = float(value_fn(G_.obs_history[boot:boot + stack].reshape(1, -1))[0]) * g ** td
            v += sum(r * g ** i for i, r in enumerate(G_.rewards[cur:cur + td]))
            assert tv[b, j] == pytest.approx(v, rel=1e-5) and tr[b, j] == G_.rewards[cur]
            assert np.allclose(tp[b, j], G_.child_visits[cur])


def make_batch_spec(games, positions, config, value_fn, weights=None, rng=None, policy_re=None, obs_dtype=np.float32):
    """hanabizero_amd.learner.make_batch written the straightforward way (reanalyze_worker.py:148-168, 249-304, 374-399 line by
    line: a Python loop per unroll step and per reward term) -- the specification the vectorised function is compared with."""
    rng = rng or np.random
    U, td, stack, A, g = config.num_unroll_steps, config.td_steps, config.stacked_observations, config.action_space_size, config.discount
    B = len(games)
    D = config.obs_shape // stack
    # (outputs are written in place: the per-sample lists + np.stack of the straightforward version cost more than
    # everything else in this function)
    obs_batch = np.empty((B, stack + U, D), obs_dtype)
    value_obs = np.zeros((B * (U + 1), config.obs_shape), obs_dtype)  # zero_obs past the end of a game
    value_mask = np.zeros(B * (U + 1), np.float64)
    action_lst, mask_lst = [], []
    k = 0
    for b, (game, pos) in enumerate(zip(games, positions)):
        acts = [int(a) for a in game.actions[pos:pos + U]]
        mask = [1.0] * len(acts) + [0.0] * (U - len(acts))
        acts += [int(rng.randint(0, A)) for _ in range(U - len(acts))]
        obs_batch[b] = game.obs(pos, extra_len=U, padding=True)
        action_lst.append(acts)
        mask_lst.append(mask)
        traj_len = len(game)
        game_obs = np.asarray(game.obs(pos + td, U))  # :204-222 bootstrap observations
        for cur in range(pos, pos + U + 1):
            if cur + td < traj_len:
                value_mask[k] = 1.0
                beg = cur - pos
                value_obs[k].reshape(stack, D)[:] = game_obs[beg:beg + stack]
            k += 1
    values = np.asarray(value_fn(value_obs), dtype=np.float64).reshape(-1) * (g ** td) * value_mask
    target_value = np.zeros((B, U + 1), np.float32)
    target_reward = np.zeros((B, U + 1), np.float32)
    target_policy = np.zeros((B, U + 1, A), np.float32)
    k = 0
    for b, (game, pos) in enumerate(zip(games, positions)):
        traj_len = len(game)
        for j, cur in enumerate(range(pos, pos + U + 1)):
            v = values[k]
            for i, r in enumerate(game.rewards[cur:cur + td]):
                v += float(r) * g ** i  # (the reference's rewards are Python floats: float64 products, whatever the history stores)
            if cur < traj_len:
                target_value[b, j], target_reward[b, j] = v, game.rewards[cur]
                target_policy[b, j] = game.child_visits[cur]
            k += 1
    if policy_re is not None and len(policy_re):
        target_policy[:len(policy_re)] = policy_re
    w = np.ones(B, np.float32) if weights is None else np.asarray(weights, np.float32)
    inputs = (obs_batch, np.asarray(action_lst, np.int64), np.asarray(mask_lst, np.float32), np.arange(B), w, np.zeros(B))
    return inputs, (target_reward[:, :U + 1], target_value, target_policy)


def test_make_batch_equals_its_line_by_line_specification():
    from hanabizero_amd.config import make_config
    from hanabizero_amd.game import GameHistory
    from hanabizero_amd.learner import make_batch
    cfg = make_config("Hanabi-Small", simulations=10, stack=2, p_mcts_num=8, batch_size=32)
    A, stack = cfg.action_space_size, cfg.stacked_observations
    D = cfg.obs_shape // stack
    rng = np.random.RandomState(5)
    games = []
    for T in (3, 7, 12, 30, 31, 9):  # (shorter than td_steps, shorter than the unroll, long)
        games.append(GameHistory.from_arrays(None, cfg, rng.randint(0, A, T), rng.rand(T).astype(np.float32),
                                             rng.dirichlet(np.ones(A), T).astype(np.float32), rng.rand(T).astype(np.float32),
                                             np.ones((T + 1, A), np.uint8), (rng.rand(T + 1, D) < 0.3).astype(np.uint8)))
    gs = [games[i % len(games)] for i in range(40)]
    pos = [int(rng.randint(0, len(g))) for g in gs]
    pos[:6] = [len(g) - 1 for g in gs[:6]]  # the last position of every game
    value_fn = lambda o: (o.astype(np.float64) * np.arange(1, o.shape[1] + 1)).sum(1) * 1e-3
    pol_re = rng.dirichlet(np.ones(A), (7, cfg.num_unroll_steps + 1)).astype(np.float32)
    for dt in (np.float32, np.uint8):
        for pr in (None, pol_re):
            a = make_batch(gs, pos, cfg, value_fn, weights=rng.rand(40), rng=np.random.RandomState(1), policy_re=pr, obs_dtype=dt)
            b = make_batch_spec(gs, pos, cfg, value_fn, weights=a[0][4], rng=np.random.RandomState(1), policy_re=pr, obs_dtype=dt)
            for x, y in zip(a[0] + a[1], b[0] + b[1]):
                x, y = np.asarray(x), np.asarray(y)
                assert x.shape == y.shape and x.dtype == y.dtype and (x == y).all()


@pytest.mark.gpu
def test_bf16_learner_step_on_gpu_and_weights_reach_the_engine():
    from hanabizero_amd.learner import make_optimizer, update_weights
    from hanabizero_amd.model import InferenceEngine
    cfg = _cfg("Hanabi-Full", stack=4)
    torch.manual_seed(0)
    net = cfg.get_uniform_network().cuda()
    opt = make_optimizer(net, cfg)
    cfg.lr_init = 0.05
    for grp in opt.param_groups:
        grp["lr"] = 0.05
    batch = _batch(cfg, 64, 5)
    losses = [update_weights(net, batch, opt, cfg, amp=torch.bfloat16)[0][1] for _ in range(10)]
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0]
    # the actor side picks the new weights up (selfplay_worker.py:177-184): BN folded, fused kernels rebuilt
    net.eval()
    eng = InferenceEngine(net.cpu(), cfg.value_support.max, dtype=torch.bfloat16, device="cuda")
    obs = torch.from_numpy(batch[0][0][:, :4].reshape(64, -1)).cuda()
    v, p, h = eng.initial(obs)
    with torch.no_grad():
        out = net.cuda().initial_inference(obs)
    assert np.allclose(v.cpu().numpy(), np.asarray(out.value).reshape(-1), atol=0.25, rtol=0.05)
    assert torch.isfinite(h.float()).all() and p.shape == (64, cfg.action_space_size)


@pytest.mark.gpu
def test_graphed_learner_step_equals_eager_steps():
    """GraphedUpdate (the whole learner step as one hipGraph replay, fused SGD with the learning rate in a device tensor)
    walks the same trajectory as update_weights: same losses, same priorities, same weights after several steps with a
    changing learning rate, within bf16-autocast noise between the fused and the foreach optimiser kernels."""
    import copy
    from hanabizero_amd.learner import GraphedUpdate, adjust_lr, make_optimizer, update_weights
    cfg = _cfg("Hanabi-Full", stack=4)
    cfg.lr_init, cfg.lr_warm_step = 0.05, 4
    torch.manual_seed(0)
    net_a = cfg.get_uniform_network().cuda()
    net_b = copy.deepcopy(net_a)
    opt_a = make_optimizer(net_a, cfg)
    opt_b = make_optimizer(net_b, cfg, capturable=True)
    step_b = GraphedUpdate(net_b, opt_b, cfg, 32)
    batches = [_batch(cfg, 32, s) for s in range(3)]
    for it in range(8):
        adjust_lr(cfg, opt_a, it + 1)
        adjust_lr(cfg, opt_b, it + 1)
        la, pa = update_weights(net_a, batches[it % 3], opt_a, cfg, amp=torch.bfloat16)
        lb, pb = step_b(batches[it % 3])
        assert np.allclose(la[:7], lb[:7], rtol=2e-2, atol=2e-3), (it, la, lb)
        assert np.allclose(pa, pb, rtol=5e-2, atol=5e-3)
    for (k, a), b in zip(net_a.state_dict().items(), net_b.state_dict().values()):
        assert torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-3), k
    assert float(opt_b.param_groups[0]["lr"]) == pytest.approx(0.05)


def test_replay_buffer_bookkeeping():
    from hanabizero_amd.replay import ReplayBuffer
    cfg = _cfg()

    class G:
        def __init__(self, T):
            self.T = T

        def __len__(self):
            return self.T
    rb = ReplayBuffer(cfg, transition_top=40, seed=1)
    rb.save_pools([(G(10), None), (G(7), np.arange(7) + 1.0)])
    assert rb.get_total_len() == 17 and rb.size() == 2 and rb.priorities[:10].tolist() == [1.0] * 10
    assert rb.priorities[10:].tolist() == list(np.arange(7) + 1.0)
    rb.save_game(G(5), True, 0)                       # new games enter at the current maximum priority
    assert rb.priorities[-5:].tolist() == [7.0] * 5
    games, pos, idx, w, mt = rb.prepare_batch_context(6, beta=0.4)
    assert len(set(idx.tolist())) == 6 and w.max() == 1.0 and all(0 <= p < len(g) for g, p in zip(games, pos))
    p = rb.priorities ** 0.6
    p /= p.sum()
    want = (22 * p[idx]) ** -0.4
    assert np.allclose(w, want / want.max(), rtol=1e-6)
    rb.update_priorities(idx, np.full(6, 0.5), mt)
    assert (rb.priorities[idx] == 0.5).all()
    for _ in range(4):
        rb.save_game(G(9), True, 0)
    dropped = rb.remove_to_fit()                       # 58 positions > 40: oldest games go, look-ups stay consistent
    assert dropped >= 1 and rb.get_total_len() == sum(len(g) for g in rb.buffer) <= 40 + 9
    gid, gpos = rb.game_look_up[0]
    assert gid - rb.base_idx == 0 and gpos == 0
    # a batch drawn BEFORE the eviction indexes positions that have moved: its write-back is dropped ...
    before = rb.priorities.copy()
    rb.update_priorities(idx, np.full(6, 123.0), mt)
    assert np.array_equal(rb.priorities, before)
    # ... a batch drawn AFTER it is applied (and keeps being applied after later evictions, each with its own epoch)
    for _ in range(2):
        games, pos, idx2, w2, mt2 = rb.prepare_batch_context(6, beta=0.4)
        rb.update_priorities(idx2, np.full(6, 0.25), mt2)
        assert (rb.priorities[idx2] == 0.25).all()
        for _ in range(5):
            rb.save_game(G(9), True, 0)
        assert rb.remove_to_fit() >= 1


@pytest.mark.gpu
def test_self_play_to_learner_loop_on_one_gpu():
    """actor -> drain_packed -> gather_packed -> ReplayBuffer.ingest_packed -> make_batch -> update_weights ->
    InferenceEngine.load -> actor keeps playing with the new weights (the loop of core/train.py + workers, Ray-free)."""
    from hanabizero_amd.dist import gather_packed
    from hanabizero_amd.learner import make_batch, make_optimizer, update_weights
    from hanabizero_amd.model import InferenceEngine
    from hanabizero_amd.replay import ReplayBuffer
    from hanabizero_amd.selfplay import SelfPlayActor
    cfg = _cfg("Hanabi-Small", stack=2)
    cfg.batch_size = 32
    torch.manual_seed(0)
    net = cfg.get_uniform_network()
    for p in net.parameters():
        if float(p.detach().abs().sum()) == 0.0:
            torch.nn.init.normal_(p, std=0.1)
    net.eval()
    eng = InferenceEngine(net, cfg.value_support.max, dtype=torch.bfloat16, device="cuda")
    actor = SelfPlayActor(cfg, eng, 128, seed=3)
    rb = ReplayBuffer(cfg)
    for step in range(30):
        actor.step()
        if step % 10 == 9:  # random-init nets lose a Hanabi-Small game in a handful of moves: drain often
            for buf, n, moves in gather_packed(actor.drain_packed(), actor.A, actor.W):
                rb.ingest_packed(buf, n, moves)
    assert rb.size() > 50 and rb.get_total_len() > 10 * cfg.batch_size
    g0 = rb.buffer[0]
    assert len(g0.child_visits) == len(g0) and abs(sum(g0.child_visits[0]) - 1.0) < 1e-6
    learner = cfg.get_uniform_network().cuda()
    learner.load_state_dict(net.state_dict())
    opt = make_optimizer(learner, cfg)
    for grp in opt.param_groups:
        grp["lr"] = 0.02
    value_fn = lambda o: eng.initial(torch.from_numpy(o).cuda())[0].float().cpu().numpy()
    losses = []
    for it in range(6):
        games, pos, idx, w, mt = rb.prepare_batch_context(cfg.batch_size, beta=0.4)
        batch = make_batch(games, pos, cfg, value_fn, weights=w, rng=np.random.RandomState(it))
        loss_data, prio = update_weights(learner, batch, opt, cfg, amp=torch.bfloat16)
        rb.update_priorities(idx, prio, mt)
        losses.append(loss_data[1])
    assert np.isfinite(losses).all()
    learner.eval()
    eng.load(learner.cpu())            # selfplay_worker.py:177-184: the actor picks the new weights up (in place: the
                                       # captured hipGraph keeps pointing at live tensors, nothing is captured again)
    before = int(actor.out_count[0].item())
    for _ in range(8):
        actor.step()
    torch.cuda.synchronize()
    assert int(actor.illegal_steps) == 0 and int(actor.out_count[0].item()) > before


@pytest.mark.gpu
def test_config5_loop_hanabi_full_5p_with_reanalyze_on_one_gpu():
    """BASELINE.json configs[4] on one GPU, Ray-free: Hanabi-Full 5 players (A = 48, D = 1385, mdp global) self-play ->
    drain_packed -> gather_packed -> ReplayBuffer.ingest_packed -> prepare_batch_context -> reanalyze (policy_re_context +
    prepare_policy_re: the second caller of the search kernels refreshes the policy targets of the reanalyzed part of the
    batch with the target model) -> make_batch -> update_weights (bf16 autocast) -> engine.load -> the graph-captured actor
    keeps playing with the new weights.  The loop of core/reanalyze_worker.py:148-204, 307-371, 402-422 + core/train.py:317-431."""
    from hanabizero_amd.dist import gather_packed
    from hanabizero_amd.learner import make_batch, make_optimizer, update_weights
    from hanabizero_amd.model import InferenceEngine
    from hanabizero_amd.reanalyze import policy_re_context, prepare_policy_re
    from hanabizero_amd.replay import ReplayBuffer
    from hanabizero_amd.selfplay import SelfPlayActor
    from hanabizero_amd.config import make_config
    cfg = make_config("Hanabi-Full-5p", simulations=20, stack=4, batch_size=64)
    assert (cfg.action_space_size, cfg.obs_dim, cfg.mdp) == (48, 1385, "global")
    torch.manual_seed(0)
    net = cfg.get_uniform_network()
    for p in net.parameters():
        if float(p.detach().abs().sum()) == 0.0:
            torch.nn.init.normal_(p, std=0.1)
    net.eval()
    eng = InferenceEngine(net, cfg.value_support.max, dtype=torch.bfloat16, device="cuda")       # the actors' model
    target = InferenceEngine(net, cfg.value_support.max, dtype=torch.bfloat16, device="cuda")    # the reanalyze workers' target model
    actor = SelfPlayActor(cfg, eng, 256, seed=4)
    rb = ReplayBuffer(cfg)
    for step in range(36):
        actor.step()
        if step % 12 == 11:
            for buf, n, moves in gather_packed(actor.drain_packed(), actor.A, actor.W):
                rb.ingest_packed(buf, n, moves)
    assert rb.size() > 40 and rb.get_total_len() > 4 * cfg.batch_size
    assert rb.buffer[0].legal_actions.shape[1] == 48 and rb.buffer[0].obs_history.shape[1] == 1385
    learner = cfg.get_uniform_network().cuda()
    learner.load_state_dict(net.state_dict())
    opt = make_optimizer(learner, cfg)
    for grp in opt.param_groups:
        grp["lr"] = 0.02
    value_fn = lambda o: target.initial(torch.from_numpy(o).cuda())[0].float().cpu().numpy()
    R = int(cfg.batch_size * 0.5)  # revisit_policy_search_rate: half the batch gets re-searched policy targets
    losses = []
    for it in range(3):
        games, pos, idx, w, mt = rb.prepare_batch_context(cfg.batch_size, beta=0.4)
        ctx = policy_re_context(cfg, games[:R], pos[:R], idx[:R])
        pol_re = prepare_policy_re(cfg, target, ctx, tie_seed=it)
        assert pol_re.shape == (R, cfg.num_unroll_steps + 1, 48)
        live = np.asarray(ctx[1]).reshape(R, -1) != 0
        assert np.allclose(pol_re.sum(-1)[live], 1.0) and (pol_re.sum(-1)[~live] == 0).all()
        batch = make_batch(games, pos, cfg, value_fn, weights=w, rng=np.random.RandomState(it), policy_re=pol_re)
        assert np.array_equal(batch[1][2][:R], pol_re.astype(np.float32))
        loss_data, prio = update_weights(learner, batch, opt, cfg, amp=torch.bfloat16)
        rb.update_priorities(idx, prio, mt)
        losses.append(loss_data[1])
    assert np.isfinite(losses).all()
    learner.eval()
    eng.load(learner.cpu())
    before = int(actor.out_count[0].item())
    for _ in range(10):
        actor.step()
    torch.cuda.synchronize()
    assert int(actor.illegal_steps) == 0 and int(actor.out_count[0].item()) > before


# ---- the fused Linear + BatchNorm + ReLU blocks of the learner (include/hz_train.h, hanabizero_amd/fused_train.py) ----------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,cols,relu,with_res", [(256, 512, True, False), (256, 1024, True, True), (64, 200, False, True), (7, 33, True, False),
                                                     (600, 96, True, True)])  # (more rows than a thread holds in registers: the re-reading passes)
def test_bn_act_kernels_against_fp32_reference(dtype, rows, cols, relu, with_res):
    """hz_bn_act_forward / hz_bn_act_backward against plain PyTorch fp32 of the same op on the same 16-bit inputs:
    batch_norm(training) -> (+ residual, through the 16-bit rounding autocast puts there) -> ReLU; outputs to one unit of the
    element format, statistics and affine gradients to fp32 summation order."""
    import ctypes as C
    from hanabizero_amd._lib import check, lib
    g = torch.Generator(device="cuda").manual_seed(rows * cols)
    x = (torch.randn(rows, cols, device="cuda", generator=g) * 1.7 + 0.3).to(dtype)
    res = torch.randn(rows, cols, device="cuda", generator=g).to(dtype) if with_res else None
    gamma = torch.rand(cols, device="cuda", generator=g) + 0.5
    beta = torch.randn(cols, device="cuda", generator=g) * 0.2
    rm, rv = torch.randn(cols, device="cuda", generator=g) * 0.1, torch.rand(cols, device="cuda", generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    dout = torch.randn(rows, cols, device="cuda", generator=g).to(dtype)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dt = {torch.bfloat16: 1, torch.float16: 2}[dtype]
    out, stats = torch.empty_like(x), torch.empty(2, cols, device="cuda")
    check(lib.hz_bn_act_forward(x.data_ptr(), cols, None if res is None else res.data_ptr(), cols, out.data_ptr(), cols, rows, cols, gamma.data_ptr(),
                                beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, stats[0].data_ptr(), stats[1].data_ptr(), int(relu), dt, st), "fwd")
    # fp32 reference
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    bn = torch.nn.functional.batch_norm(xr, rm_ref, rv_ref, gr, br, training=True, momentum=0.1, eps=1e-5)
    rr = res.float().requires_grad_(True) if with_res else None
    pre = bn.to(dtype).float() + rr if with_res else bn      # (autocast materialises bn's output in 16 bits before the add)
    if with_res:
        pre = bn + (bn.to(dtype).float() - bn).detach() + rr   # same value, gradient of the identity (straight through the rounding)
    want = torch.relu(pre) if relu else pre
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert torch.allclose(out.float(), want.to(dtype).float(), rtol=2 * ulp, atol=2 * ulp)
    assert torch.allclose(rm, rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv_ref, rtol=1e-5, atol=1e-6)
    assert torch.allclose(stats[0], x.float().mean(0), rtol=1e-5, atol=1e-6)
    assert torch.allclose(stats[1], torch.rsqrt(x.float().var(0, unbiased=False) + 1e-5), rtol=1e-5)
    # backward: the ReLU mask comes from the kernel's own 16-bit output
    dx, dres = torch.empty_like(x), (torch.empty_like(x) if with_res else None)
    dgamma, dbeta = torch.full((cols,), 0.5, device="cuda"), torch.full((cols,), -0.25, device="cuda")  # (accumulated into)
    check(lib.hz_bn_act_backward(dout.data_ptr(), cols, out.data_ptr(), cols, x.data_ptr(), cols, dx.data_ptr(), cols,
                                 None if dres is None else dres.data_ptr(), cols, rows, cols, gamma.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                                 dgamma.data_ptr(), dbeta.data_ptr(), int(relu), dt, st), "bwd")
    mask = (out.float() > 0).float() if relu else torch.ones_like(want)
    (pre * (dout.float() * mask).detach()).sum().backward()
    scale = float(xr.grad.abs().max())
    assert torch.allclose(dx.float(), xr.grad, rtol=4 * ulp, atol=4 * ulp * scale)
    assert torch.allclose(dgamma - 0.5, gr.grad, rtol=1e-4, atol=1e-4 * float(gr.grad.abs().max()))
    assert torch.allclose(dbeta + 0.25, br.grad, rtol=1e-4, atol=1e-4 * float(br.grad.abs().max()))
    if with_res:
        assert torch.equal(dres.float(), (dout.float() * mask).to(dtype).float())


@pytest.mark.gpu
@pytest.mark.parametrize("game,stack", [("Hanabi-Small", 2), ("Hanabi-Full-5p", 4)])
def test_fused_learner_step_matches_the_autocast_step(game, stack):
    """update_weights through fused_train.FusedTrainNet (GEMM + one launch per block) against update_weights through the module's
    own forward under bf16 autocast -- the path the reference-recorded fixtures pin (tests/test_reference_callers.py) -- from the
    same weights on the same batch: losses, priorities, every parameter's gradient direction, the stepped weights, the
    BatchNorm running statistics and counters."""
    import copy
    from hanabizero_amd.fused_train import FusedTrainNet
    from hanabizero_amd.learner import make_optimizer, update_weights
    cfg = _cfg(game, stack)
    for seed in (3, 4):  # (one step each from fresh, identical weights: after a step the two 16-bit computations have drifted apart
        torch.manual_seed(seed)  # by their own rounding, and the deep layers' gradients are sensitive to that)
        net = cfg.get_uniform_network()
        for p in net.parameters():
            if float(p.detach().abs().sum()) == 0.0:
                torch.nn.init.normal_(p, std=0.05)
        net = net.cuda()
        net2, net3 = copy.deepcopy(net), copy.deepcopy(net)
        batch = _batch(cfg, 64, seed)
        opt1, opt2 = make_optimizer(net, cfg), make_optimizer(net2, cfg)
        fused = FusedTrainNet(net2, unroll_steps=cfg.num_unroll_steps)
        update_weights(net3, batch, make_optimizer(net3, cfg), cfg, amp=None)   # the same step in fp32: what both 16-bit paths approximate
        g0 = [p.grad.clone() for p in net3.parameters()]
        l1, p1 = update_weights(net, batch, opt1, cfg, amp=torch.bfloat16)
        g1 = [p.grad.clone() for p in net.parameters()]
        l2, p2 = update_weights(fused, batch, opt2, cfg, amp=torch.bfloat16)
        g2 = [p.grad.clone() for p in net2.parameters()]
        assert np.allclose(l1, l2, rtol=2e-2, atol=1e-3), (seed, l1, l2)
        assert np.allclose(p1, p2, rtol=5e-2, atol=5e-2)
        names = [n for n, _ in net.named_parameters()]
        cosine = lambda u, v: float((u * v).sum() / (u.norm() * v.norm() + 1e-30))
        for n, z, a, b in zip(names, g0, g1, g2):
            if float(b.abs().sum()) == 0.0 and float(z.norm()) < 1e-4 * max(1.0, float(a.norm())) + 1e-5:
                continue  # a Linear bias in front of a BatchNorm: zero in fp32 too (rounding noise under autocast, exactly zero in the fused block)
            if float(z.norm()) < 1e-6:
                continue
            # every parameter's gradient: as close to the fp32 gradient as the autocast step's is (the deepest layers' gradients pass
            # through every rounding of the unrolled net: 0.90 - 0.97 against fp32 on either 16-bit path), and close to the autocast one
            c_auto, c_fused = cosine(z, a), cosine(z, b)
            assert c_fused > c_auto - 0.03 and c_fused > 0.85, (seed, n, c_auto, c_fused)
            assert cosine(a, b) > 0.9, (seed, n, cosine(a, b))
            assert 0.9 < float(b.norm() / a.norm()) < 1.1, (seed, n)
        for (n, a), b in zip(net.state_dict().items(), net2.state_dict().values()):
            if n.endswith("num_batches_tracked"):
                assert int(a) == int(b), n
            else:
                assert torch.allclose(a, b, rtol=2e-2, atol=2e-3), (n, float((a - b).abs().max()))
        # the fused model's 16-bit weight copies follow the step
        for blk in fused._blocks:
            assert torch.equal(blk.w16, blk.lin.weight.detach().to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V,smin,A,with_reward", [(201, -100, 48, True), (51, -25, 11, True), (201, -100, 20, False)])
def test_head_losses_kernel_against_fp32_reference(dtype, V, smin, A, with_reward):
    """hz_muzero_head_losses against the plain PyTorch fp32 restatement of one inference's losses (scalar_transform -> phi two-hot ->
    -(log_softmax . target), the policy cross-entropy, the weighted total) and autograd's gradients with respect to the logits; the
    heads' scalar predictions against inverse_scalar_transform."""
    from hanabizero_amd.fused_train import _HeadLosses
    from hanabizero_amd.learner import phi, scalar_transform
    from hanabizero_amd.model import inverse_scalar_transform
    import types
    B = 133
    g = torch.Generator(device="cuda").manual_seed(V + A)
    mk = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 2).to(dtype)
    value, reward, policy = mk(B, V), (mk(B, V) if with_reward else None), mk(B, A)
    tv = torch.rand(B, 3, device="cuda", generator=g) * 60 - 20     # (strided rows: column 1 is the target)
    tv[0, 1], tv[1, 1], tv[2, 1] = 3.0, 0.0, -1e6                    # an integer after the transform? (0 is), clamped far below the support
    tr = (torch.randint(-3, 4, (B, 2), device="cuda", generator=g)).float()
    tp = torch.rand(B, 2, A, device="cuda", generator=g)
    tp = tp / tp.sum(-1, keepdim=True)
    tp[5:9, 1] = 0.0                                                  # positions past the end of their game: no policy target
    weights = torch.rand(B, device="cuda", generator=g) + 0.5
    support = types.SimpleNamespace(min=smin, max=smin + V - 1, size=V)
    coeffs = (0.25, 1.0, 1.0)
    v_in, p_in = value.clone().requires_grad_(True), policy.clone().requires_grad_(True)
    r_in = reward.clone().requires_grad_(True) if with_reward else None
    tot, L, P = _HeadLosses.apply(v_in, r_in, p_in, tv[:, 1], tr[:, 0] if with_reward else None, tp[:, 1], weights, support, coeffs)
    tot.sum().backward()
    # fp32 reference
    vr, pr = value.float().requires_grad_(True), policy.float().requires_grad_(True)
    rr = reward.float().requires_grad_(True) if with_reward else None
    ce = lambda logits, target: -(torch.log_softmax(logits, 1) * target).sum(1)
    vphi = phi(scalar_transform(tv[:, 1:2].clone()), smin, smin + V - 1, V)[:, 0]
    vl = ce(vr, vphi)
    pl = ce(pr, tp[:, 1])
    rl = ce(rr, phi(scalar_transform(tr[:, 0:1].clone()), smin, smin + V - 1, V)[:, 0]) if with_reward else torch.zeros(B, device="cuda")
    want = weights / B * (coeffs[2] * pl + coeffs[0] * vl + coeffs[1] * rl)
    want.sum().backward()
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-4, atol=1e-4)
    assert torch.allclose(L[:, 0], pl, **tol) and torch.allclose(L[:, 1], vl, **tol) and torch.allclose(L[:, 2], rl, **tol)
    assert torch.allclose(tot, want, **tol) and torch.equal(tot, L[:, 3])
    ulp = {torch.float32: 1e-6, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    for got, ref in ((v_in.grad, vr.grad), (p_in.grad, pr.grad)) + (((r_in.grad, rr.grad),) if with_reward else ()):
        assert got.dtype == dtype and torch.allclose(got.float(), ref, rtol=4 * ulp, atol=4 * ulp * float(ref.abs().max()))
    assert torch.allclose(P[:, 0], inverse_scalar_transform(value.float(), smin, smin + V - 1).reshape(-1), rtol=1e-4, atol=1e-4)
    if with_reward:
        assert torch.allclose(P[:, 1], inverse_scalar_transform(reward.float(), smin, smin + V - 1).reshape(-1), rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,groups,cols,relu,with_res", [(256, 6, 256, True, False), (256, 5, 512, True, True), (37, 3, 201, False, True),
                                                            (300, 2, 40, True, False)])
def test_grouped_bn_act_kernels_equal_one_call_per_group(dtype, rows, groups, cols, relu, with_res):
    """hz_bn_act_forward_groups / _backward_groups over `groups` stacked batches (+ hz_bn_groups_finish for what crosses the groups)
    == one hz_bn_act_forward / _backward call per batch, in order, on the same running statistics and gradient accumulators:
    outputs, saved statistics, running statistics, input / residual gradients and the accumulated affine gradients, bit for bit
    (twice in a row)."""
    import ctypes as C
    from hanabizero_amd._lib import BnFinish, check, lib
    g = torch.Generator(device="cuda").manual_seed(rows * cols + groups)
    R = rows * groups
    x = (torch.randn(R, cols, device="cuda", generator=g) * 1.7 + 0.3).to(dtype)
    res = torch.randn(R, cols, device="cuda", generator=g).to(dtype) if with_res else None
    gamma = torch.rand(cols, device="cuda", generator=g) + 0.5
    beta = torch.randn(cols, device="cuda", generator=g) * 0.2
    dout = torch.randn(R, cols, device="cuda", generator=g).to(dtype)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dt = {torch.bfloat16: 1, torch.float16: 2}[dtype]
    ptr = lambda t: None if t is None else t.data_ptr()
    scratch = torch.full((groups, 2, cols), float("nan"), device="cuda")
    def finish(dst0, dst1, backward):
        e = (BnFinish * 1)(BnFinish(dst0=dst0.data_ptr(), dst1=dst1.data_ptr(), scratch=scratch.data_ptr(), cols=cols, groups=groups, momentum=0.1))
        table = torch.frombuffer(bytearray(bytes(e)), dtype=torch.uint8).cuda()
        check(lib.hz_bn_groups_finish(table.data_ptr(), 1, cols, int(backward), st), "finish")
        torch.cuda.synchronize()
    rm0, rv0 = torch.randn(cols, device="cuda", generator=g) * 0.1, torch.rand(cols, device="cuda", generator=g) + 0.5
    rm_a, rv_a, rm_b, rv_b = rm0.clone(), rv0.clone(), rm0.clone(), rv0.clone()
    dg_a, db_a = torch.full((cols,), 0.5, device="cuda"), torch.full((cols,), -0.25, device="cuda")
    dg_b, db_b = dg_a.clone(), db_a.clone()
    for _ in range(2):
        out_a, stats_a = torch.empty_like(x), torch.empty(2, groups, cols, device="cuda")
        out_b, stats_b = torch.empty_like(x), torch.empty(2, groups, cols, device="cuda")
        check(lib.hz_bn_act_forward_groups(x.data_ptr(), cols, ptr(res), cols, out_a.data_ptr(), cols, rows, groups, cols, gamma.data_ptr(), beta.data_ptr(),
                                           rm_a.data_ptr(), rv_a.data_ptr(), 0.1, 1e-5, stats_a[0].data_ptr(), stats_a[1].data_ptr(), scratch.data_ptr(),
                                           int(relu), dt, st), "fwd groups")
        finish(rm_a, rv_a, False)
        for k in range(groups):
            s = slice(k * rows, (k + 1) * rows)
            check(lib.hz_bn_act_forward(x[s].data_ptr(), cols, ptr(None if res is None else res[s]), cols, out_b[s].data_ptr(), cols, rows, cols,
                                        gamma.data_ptr(), beta.data_ptr(), rm_b.data_ptr(), rv_b.data_ptr(), 0.1, 1e-5, stats_b[0, k].data_ptr(),
                                        stats_b[1, k].data_ptr(), int(relu), dt, st), "fwd")
        assert torch.equal(out_a.view(torch.int16), out_b.view(torch.int16)) and torch.equal(stats_a, stats_b)
        assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)
        dx_a, dx_b = torch.empty_like(x), torch.empty_like(x)
        dr_a, dr_b = (torch.empty_like(x), torch.empty_like(x)) if with_res else (None, None)
        check(lib.hz_bn_act_backward_groups(dout.data_ptr(), cols, out_a.data_ptr(), cols, x.data_ptr(), cols, dx_a.data_ptr(), cols, ptr(dr_a), cols, rows,
                                            groups, cols, gamma.data_ptr(), stats_a[0].data_ptr(), stats_a[1].data_ptr(), dg_a.data_ptr(), db_a.data_ptr(),
                                            scratch.data_ptr(), int(relu), dt, st), "bwd groups")
        finish(db_a, dg_a, True)
        acc_g, acc_b = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
        for k in range(groups):   # (the grouped kernel adds the groups' sums up first, then adds the total to the accumulator)
            s = slice(k * rows, (k + 1) * rows)
            one_g, one_b = torch.zeros(cols, device="cuda"), torch.zeros(cols, device="cuda")
            check(lib.hz_bn_act_backward(dout[s].data_ptr(), cols, out_b[s].data_ptr(), cols, x[s].data_ptr(), cols, dx_b[s].data_ptr(), cols,
                                         ptr(None if dr_b is None else dr_b[s]), cols, rows, cols, gamma.data_ptr(), stats_b[0, k].data_ptr(),
                                         stats_b[1, k].data_ptr(), one_g.data_ptr(), one_b.data_ptr(), int(relu), dt, st), "bwd")
            acc_g, acc_b = acc_g + one_g, acc_b + one_b
        dg_b, db_b = dg_b + acc_g, db_b + acc_b
        assert torch.equal(dx_a.view(torch.int16), dx_b.view(torch.int16))
        assert torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b)
        if with_res:
            assert torch.equal(dr_a.view(torch.int16), dr_b.view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_unrolled_losses_equal_one_launch_per_inference(dtype):
    """hz_muzero_unrolled_losses (_UnrolledLosses: every inference of the unrolled step in one launch, targets read through their
    strides) == hz_muzero_head_losses per inference (_HeadLosses) on the same logits: losses, predictions, row totals and the
    gradients of all three heads' logits, bit for bit."""
    import types
    from hanabizero_amd.fused_train import _HeadLosses, _UnrolledLosses
    B, U, V, smin, A = 77, 5, 201, -100, 48
    g = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 2).to(dtype)
    value, reward, policy = mk((U + 1) * B, V), mk(U * B, V), mk((U + 1) * B, A)
    tv = torch.rand(B, U + 1, device="cuda", generator=g) * 60 - 20
    tr = torch.randint(-3, 4, (B, U), device="cuda", generator=g).float()
    tp = torch.rand(B, U + 1, A, device="cuda", generator=g)
    tp = tp / tp.sum(-1, keepdim=True)
    tp[5:9, 3:] = 0.0
    weights = torch.rand(B, device="cuda", generator=g) + 0.5
    support = types.SimpleNamespace(min=smin, max=smin + V - 1, size=V)
    coeffs = (0.25, 1.0, 1.0)
    va, ra, pa = (t.clone().requires_grad_(True) for t in (value, reward, policy))
    tot, L, P = _UnrolledLosses.apply(va, ra, pa, tv, tr, tp, weights, support, coeffs)
    up = torch.rand((U + 1) * B, device="cuda", generator=g)   # (an upstream gradient that differs per row)
    (tot * up).sum().backward()
    vb, rb, pb = (t.clone().requires_grad_(True) for t in (value, reward, policy))
    tots, Ls, Ps = [], [], []
    for k in range(U + 1):
        s = slice(k * B, (k + 1) * B)
        t, l, p = _HeadLosses.apply(vb[s], rb[(k - 1) * B:k * B] if k else None, pb[s], tv[:, k], tr[:, k - 1] if k else None, tp[:, k], weights,
                                    support, coeffs)
        tots.append(t), Ls.append(l), Ps.append(p)
    (torch.cat(tots) * up).sum().backward()
    assert torch.equal(tot, torch.cat(tots)) and torch.equal(L, torch.cat(Ls)) and torch.equal(P, torch.cat(Ps))
    for a, b in ((va, vb), (ra, rb), (pa, pb)):   # (as numbers: adding the slices' zero-filled gradients up turns a -0 into +0)
        assert torch.equal(a.grad, b.grad) and not bool(torch.isnan(a.grad).any())


@pytest.mark.gpu
@pytest.mark.parametrize("game,stack", [("Hanabi-Small", 2), ("Hanabi-Full-5p", 4)])
def test_stacked_heads_equal_the_heads_inference_by_inference(game, stack):
    """FusedTrainNet.compute_losses (the three heads once over the stacked hidden states of all inferences, every BatchNorm batch
    with its own statistics) against compute_losses_stepwise (initial_inference / recurrent_inference as the module is called):
    the same function of the same weights -- losses and priorities to 16-bit GEMM rounding (another row count, another summation
    order inside the library), gradients in the same direction, the same running statistics and counters."""
    import copy
    from hanabizero_amd.fused_train import FusedTrainNet
    from hanabizero_amd.learner import make_optimizer, update_weights
    cfg = _cfg(game, stack)
    torch.manual_seed(11)
    net = cfg.get_uniform_network()
    for p in net.parameters():
        if float(p.detach().abs().sum()) == 0.0:
            torch.nn.init.normal_(p, std=0.05)
    net = net.cuda()
    net2 = copy.deepcopy(net)
    batch = _batch(cfg, 64, 11)
    a, b = FusedTrainNet(net, unroll_steps=cfg.num_unroll_steps), FusedTrainNet(net2, unroll_steps=cfg.num_unroll_steps)
    b.compute_losses = b.compute_losses_stepwise
    l1, p1 = update_weights(a, batch, make_optimizer(net, cfg), cfg, amp=torch.bfloat16)
    l2, p2 = update_weights(b, batch, make_optimizer(net2, cfg), cfg, amp=torch.bfloat16)
    assert np.allclose(l1, l2, rtol=5e-3, atol=5e-4), (l1, l2)
    assert np.allclose(p1, p2, rtol=2e-2, atol=2e-2)
    cosine = lambda u, v: float((u * v).sum() / (u.norm() * v.norm() + 1e-30))
    for (n, u), v in zip(net.named_parameters(), net2.parameters()):
        if float(u.grad.norm()) < 1e-6 and float(v.grad.norm()) < 1e-6:
            continue
        assert cosine(u.grad, v.grad) > 0.98 and 0.95 < float(u.grad.norm() / v.grad.norm()) < 1.05, (n, cosine(u.grad, v.grad))
    for (n, u), v in zip(net.state_dict().items(), net2.state_dict().values()):
        if n.endswith("num_batches_tracked"):
            assert int(u) == int(v), n
        else:
            assert torch.allclose(u, v, rtol=5e-3, atol=5e-4), (n, float((u - v).abs().max()))
