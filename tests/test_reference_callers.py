"""The Python callers either side of the hot path against fixtures recorded from the REFERENCE's own Python
(tools/gen_golden_callers.py imports /root/reference/core/{utils,game,selfplay_worker,train,reanalyze_worker}.py in the
authoring container): select_action, GameHistory + put(), one update_weights step, make_batch inputs / targets.
CPU tests pin the host logic in fp32 / fp64; the gpu tests run the HIP select_action kernel and the learner step on the card."""
import os

import numpy as np
import pytest
import torch

from tests.netgold import fill_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fx(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


# ------------------------------------------------------------------------------------------------ select_action
def _restated_select_action(counts, legal, u, temperature, deterministic):
    """tests/restate.py's restatement (what the actor tests replay moves with), here checked against the reference."""
    from tests.restate import ref_select_action
    a, ent, masked = ref_select_action(counts, legal, u, temperature, deterministic)
    return a, ent


def test_select_action_restatement_equals_reference():
    fx = _fx("select_action.npz")
    for i in range(len(fx["action"])):
        A = int(fx["num_actions"][i])
        a, ent = _restated_select_action(fx["counts"][i, :A], fx["legal"][i, :A], float(fx["uniform"][i]),
                                         float(fx["temperature"][i]), bool(fx["deterministic"][i]))
        assert a == int(fx["action"][i]), i
        assert abs(ent - float(fx["entropy"][i])) < 1e-12, i


@pytest.mark.gpu
def test_hip_select_action_equals_reference():
    """hz_select_action (include/hz_selfplay.h) on the reference's recorded cases: same action for the uniform numpy drew,
    same entropy (fp64), counts masked in place as core/utils.py:282-284 does."""
    import ctypes as C
    from hanabizero_amd._lib import check, lib
    fx = _fx("select_action.npz")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for A in (11, 20, 48):
        for det in (False, True):
            for T in sorted(set(fx["temperature"].tolist())):
                sel = np.nonzero((fx["num_actions"] == A) & (fx["deterministic"] == det) & (fx["temperature"] == T))[0]
                if len(sel) == 0:
                    continue
                counts = torch.from_numpy(fx["counts"][sel, :A].astype(np.int32)).cuda().contiguous()
                legal = torch.from_numpy(fx["legal"][sel, :A].astype(np.uint8)).cuda().contiguous()
                u = torch.from_numpy(fx["uniform"][sel].astype(np.float64)).cuda()
                act = torch.full((len(sel),), -7, dtype=torch.int32, device="cuda")
                ent = torch.zeros(len(sel), dtype=torch.float64, device="cuda")
                check(lib.hz_select_action(len(sel), A, counts.data_ptr(), legal.data_ptr(), u.data_ptr(), float(T), int(det),
                                           act.data_ptr(), ent.data_ptr(), st), "hz_select_action")
                torch.cuda.synchronize()
                assert act.cpu().numpy().tolist() == fx["action"][sel].tolist(), (A, det, T)
                assert np.allclose(ent.cpu().numpy(), fx["entropy"][sel], rtol=0, atol=1e-12), (A, det, T)
                masked = fx["counts"][sel, :A] * fx["legal"][sel, :A]
                assert (counts.cpu().numpy() == masked).all()


# ------------------------------------------------------------------------------------------------ GameHistory + put
def test_game_history_and_put_equal_reference():
    """The array-backed GameHistory driven move by move exactly as the reference's (core/game.py:49-214), then put()
    (selfplay_worker.py:29-39): every field the reference's save_file() reports, its obs() windows and step_obs()."""
    from hanabizero_amd.config import make_config
    from hanabizero_amd.game import GameHistory, reshape_turn_rewards
    fx = _fx("game_history.npz")
    cfg = make_config("Hanabi-Small", stack=int(fx["stack"]))
    for g in range(3):
        I = lambda k: fx["g%d_in_%s" % (g, k)]
        O = lambda k: fx["g%d_out_%s" % (g, k)]
        T = int(fx["g%d_len" % g])
        gh = GameHistory(None, max_length=cfg.max_moves, config=cfg)
        gh.init([I("obs")[0] for _ in range(cfg.stacked_observations)], I("legal")[0])
        for t in range(T):
            gh.store_search_stats(list(I("visits")[t]), float(I("value")[t]))
            gh.append(int(I("action")[t]), I("obs")[t + 1], int(I("reward")[t]), I("legal")[t + 1])
        assert np.array_equal(np.asarray(gh.step_obs()), fx["g%d_step_obs" % g])
        gh.game_over()
        reshape_turn_rewards(gh)
        saved = gh.save_file()
        assert len(gh) == T
        for k in ("vis", "root", "a", "o", "r", "la"):
            assert saved[k].shape == O(k).shape and np.array_equal(saved[k], O(k)), (g, k)
        assert saved["vis"].dtype == np.float64 and saved["root"].dtype == np.float64
        assert np.array_equal(np.asarray(gh.obs(min(1, T), extra_len=2, padding=True)), fx["g%d_obs_1_2_pad" % g])
        assert np.array_equal(np.asarray(gh.obs(T, extra_len=5, padding=True)), fx["g%d_obs_last_5_pad" % g])
        assert np.array_equal(np.asarray(gh.zero_obs()), fx["g%d_zero_obs" % g])
        # ... and the whole-array constructor the GPU records go through leaves the same object behind
        raw_r = I("reward")
        vis = I("visits") / I("visits").sum(1, keepdims=True)
        g2 = GameHistory.from_arrays(None, cfg, I("action"), raw_r.copy(), vis, I("value").astype(np.float64), I("legal"), I("obs"))
        reshape_turn_rewards(g2)
        for k in ("vis", "root", "a", "o", "r", "la"):
            assert np.array_equal(g2.save_file()[k], O(k)), (g, k)


# ------------------------------------------------------------------------------------------------ learner step
def _learner_case(game, device, amp=None, fused=False):
    from hanabizero_amd.config import make_config
    from hanabizero_amd.learner import make_optimizer, update_weights
    fx = _fx("learner_step_%s.npz" % game)
    cfg = make_config(game, stack=int(fx["stack"]), batch_size=int(fx["obs"].shape[0]), lr=float(fx["lr"]))
    for k in ("momentum", "weight_decay", "max_grad_norm", "value_loss_coeff", "priority_reward_ratio", "prioritized_replay_eps"):
        assert float(getattr(cfg, k)) == pytest.approx(float(fx[k])), k  # the reference config's hyper-parameters
    assert (cfg.num_unroll_steps, cfg.action_space_size, cfg.obs_dim) == (int(fx["U"]), int(fx["A"]), int(fx["D"]))
    torch.manual_seed(0)
    net = cfg.get_uniform_network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
    net = net.to(device)
    assert [n for n, _ in net.named_parameters()] == [str(n) for n in fx["param_names"]]
    if fused:  # (the learner's hand-written blocks and losses around the same parameters: hanabizero_amd/fused_train.py)
        from hanabizero_amd.fused_train import FusedTrainNet
        net = FusedTrainNet(net, unroll_steps=cfg.num_unroll_steps)
    opt = make_optimizer(net, cfg)
    batch = ((fx["obs"].astype(np.float32), fx["action"], fx["mask"], fx["indices"], fx["weights"], np.zeros(len(fx["weights"]))),
             (fx["target_reward"], fx["target_value"], fx["target_policy"]))
    out = []
    for it in range(2):
        loss_data, prio = update_weights(net, batch, opt, cfg, amp=amp)
        grads = np.array([float(p.grad.double().norm()) for p in net.parameters()])
        out.append((np.array(loss_data, np.float64), np.asarray(prio, np.float64), grads))
    return fx, net, out


def _digest(t):
    t = t.detach().double().reshape(-1).cpu()
    return np.array([float(t.sum()), float(t.abs().sum()), float((t * t).sum())] + [float(x) for x in t[:5]] +
                    [0.0] * max(0, 5 - t.numel()))


@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
def test_update_weights_equals_reference_step_fp32(game):
    """learner.update_weights in fp32 on the CPU against two consecutive steps of the reference's update_weights
    (core/train.py:59-314, amp_type 'none') from the same state_dict on the same batch: losses, priorities, per-parameter
    gradient norms after clipping, and digests (sum, |sum|, sum of squares, first elements) of every updated parameter and
    BatchNorm running statistic."""
    fx, net, out = _learner_case(game, "cpu")
    for it, (loss, prio, grads) in enumerate(out):
        assert np.allclose(loss[[0, 1, 2, 4, 5, 6]], fx["loss_data_%d" % it][[0, 1, 2, 4, 5, 6]], rtol=2e-5, atol=1e-6), (it, loss)
        assert np.allclose(prio, fx["priority_%d" % it], rtol=1e-4, atol=1e-4), it
        assert np.allclose(grads, fx["grad_norm_%d" % it], rtol=2e-3, atol=1e-7), it
    want = fx["param_digest_1"]
    got = np.stack([_digest(p) for _, p in net.named_parameters()])
    assert np.allclose(got, want, rtol=1e-4, atol=2e-5)
    bufs = [(n, b) for n, b in net.named_buffers() if b.dtype.is_floating_point]
    assert [n for n, _ in bufs] == [str(n) for n in fx["buffer_names"]]
    assert np.allclose(np.stack([_digest(b) for _, b in bufs]), fx["buffer_digest_1"], rtol=1e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
def test_update_weights_on_gpu_equals_reference_step(game):
    """The same two steps on the MI355X in fp32 (hipBLASLt GEMMs sum in another order; the train-mode BatchNorm over a batch
    of 4-8 samples amplifies that into the second step: first step tight, second step looser), and under the bf16 autocast
    the product trains with -- losses within bf16's accuracy of the reference's fp32 step."""
    fx, net, out = _learner_case(game, "cuda")
    for it, (loss, prio, grads) in enumerate(out):
        tol = 1e-3 if it == 0 else 2e-2
        assert np.allclose(loss[[0, 1, 2, 4, 5, 6]], fx["loss_data_%d" % it][[0, 1, 2, 4, 5, 6]], rtol=tol, atol=1e-4), (it, loss)
        assert np.allclose(prio, fx["priority_%d" % it], rtol=20 * tol, atol=20 * tol), it
        if it == 0:
            assert np.allclose(grads, fx["grad_norm_%d" % it], rtol=2e-2, atol=1e-6), it
    got = np.stack([_digest(p) for _, p in net.named_parameters()])
    want = fx["param_digest_1"]
    assert np.allclose(got[:, 1:3], want[:, 1:3], rtol=2e-3, atol=1e-3)                # sum |w|, sum w^2
    assert (np.abs(got[:, 0] - want[:, 0]) <= 1e-4 * want[:, 1] + 1e-3).all()          # sum w: cancels, so against sum |w|
    fx, net, out = _learner_case(game, "cuda", amp=torch.bfloat16)
    for it, (loss, prio, grads) in enumerate(out):
        assert np.allclose(loss[[0, 1, 2, 4, 5, 6]], fx["loss_data_%d" % it][[0, 1, 2, 4, 5, 6]], rtol=3e-2, atol=3e-2), (it, loss)
    # ... and through the hand-written blocks and losses (fused_train.FusedTrainNet: what the configs[4] loop trains with): the
    # same bf16 bar against the reference's recorded fp32 step, priorities and first-step gradient norms included
    fx, net, out = _learner_case(game, "cuda", amp=torch.bfloat16, fused=True)
    for it, (loss, prio, grads) in enumerate(out):
        tol = 3e-2 if it == 0 else 0.15   # (second step: weights that differ by one 16-bit step of rounding, 4 - 8 samples per BatchNorm)
        assert np.allclose(loss[[0, 1, 2, 4, 5, 6]], fx["loss_data_%d" % it][[0, 1, 2, 4, 5, 6]], rtol=tol, atol=tol), (it, loss, fx["loss_data_%d" % it])
        if it == 0:  # (the second step of a batch of 4-8 samples through train-mode BatchNorm in 16 bits: losses hold, per-sample values scatter)
            ref = fx["priority_%d" % it]   # |predicted scalar - target| of 4 - 8 samples normalised together: 16-bit noise of a few % of the largest
            assert np.allclose(prio, ref, rtol=0.15, atol=0.05 * float(ref.max())), (it, prio, ref)
            names = [str(n) for n in fx["param_names"]]
            big = np.array([fx["grad_norm_0"][k] > 1e-3 for k in range(len(names))])   # (Linear biases in front of a BatchNorm: zero in exact arithmetic)
            ratio = grads[big] / fx["grad_norm_0"][big]
            worst = sorted(zip(np.abs(np.log(ratio)), np.array(names)[big], ratio), reverse=True)[:6]
            assert np.allclose(grads[big], fx["grad_norm_0"][big], rtol=0.35), worst   # (16-bit, 4 - 8 samples per BatchNorm: magnitudes, not digits)


# ------------------------------------------------------------------------------------------------ batch inputs / targets
def test_make_batch_equals_reference_workers():
    """learner.make_batch against BatchWorker_CPU.make_batch + BatchWorker_GPU._prepare_reward_value / _prepare_policy_non_re
    (core/reanalyze_worker.py:148-204, 249-304, 374-399) on the same games and positions, the same net as target model."""
    from hanabizero_amd.config import make_config
    from hanabizero_amd.game import GameHistory
    from hanabizero_amd.learner import make_batch
    fx = _fx("batch_targets_Hanabi-Small.npz")
    cfg = make_config("Hanabi-Small", stack=int(fx["stack"]), batch_size=len(fx["pick"]))
    assert (cfg.td_steps, cfg.num_unroll_steps, float(cfg.discount)) == (int(fx["td_steps"]), int(fx["U"]), float(fx["discount"]))
    games = []
    for i in range(4):
        I = lambda k: fx["game%d_%s" % (i, k)]
        vis = I("visits") / I("visits").sum(1, keepdims=True)
        games.append(GameHistory.from_arrays(None, cfg, I("action"), I("reward"), vis, I("value").astype(np.float64), I("legal"), I("obs")))
    torch.manual_seed(0)
    net = cfg.get_uniform_network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
    net.eval()

    def value_fn(obs):
        with torch.no_grad():
            return np.asarray(net.initial_inference(torch.from_numpy(obs).float()).value).reshape(-1)
    game_lst = [games[i] for i in fx["pick"]]
    np.random.seed(int(fx["pad_action_seed"]))  # the reference pads action windows with np.random.randint (:160)
    (obs, action, mask, idx, w, mt), (t_reward, t_value, t_policy) = make_batch(game_lst, fx["positions"].tolist(), cfg, value_fn,
                                                                                weights=fx["weights"])
    assert np.array_equal(obs, fx["in_obs"].astype(np.float32))
    assert np.array_equal(action, fx["in_action"]) and np.array_equal(mask, fx["in_mask"])
    assert np.allclose(w, fx["weights"])
    assert np.allclose(t_reward, fx["target_reward"], atol=1e-6)
    assert np.allclose(t_value, fx["target_value"], rtol=1e-5, atol=1e-4)
    assert np.allclose(t_policy, fx["target_policy"], atol=1e-7)


def test_policy_re_context_equals_reference_worker():
    """reanalyze.policy_re_context against BatchWorker_CPU._prepare_policy_re_context (reanalyze_worker.py:101-144): the
    observation windows, masks and legal-action rows the refreshed-policy search is prepared from."""
    from hanabizero_amd.config import make_config
    from hanabizero_amd.game import GameHistory
    from hanabizero_amd.reanalyze import policy_re_context
    fx = _fx("batch_targets_Hanabi-Small.npz")
    cfg = make_config("Hanabi-Small", stack=int(fx["stack"]))
    games = []
    for i in range(4):
        I = lambda k: fx["game%d_%s" % (i, k)]
        vis = I("visits") / I("visits").sum(1, keepdims=True)
        games.append(GameHistory.from_arrays(None, cfg, I("action"), I("reward"), vis, I("value").astype(np.float64), I("legal"), I("obs")))
    R = int(fx["re_num"])
    game_lst = [games[i] for i in fx["pick"][:R]]
    obs, mask, state_index, indices, child_visits, traj_lens, legal = policy_re_context(cfg, game_lst, fx["positions"][:R].tolist())
    assert np.array_equal(obs, fx["re_obs"].astype(np.float32)) and list(mask) == fx["re_mask"].tolist()
    assert list(state_index) == fx["re_state_index"].tolist() and list(indices) == fx["re_indices"].tolist()
    assert list(traj_lens) == fx["re_traj_lens"].tolist() and np.array_equal(np.asarray(legal), fx["re_legal"])
    assert all(np.array_equal(a, g.child_visits) for a, g in zip(child_visits, game_lst))
