"""GPU: the two other callers of the search kernels against oracle-driven replays."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine(game, sims, stack, dtype=torch.float32):
    from hanabizero_amd.config import make_config
    from hanabizero_amd.model import InferenceEngine
    from tests.netgold import fill_state_dict
    cfg = make_config(game, simulations=sims, stack=stack)
    net = cfg.get_uniform_network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
    net.eval()
    if dtype == "fp16x2":  # the fp32 engine whose recurrent inference is the MFMA kernel's fp16-pair build (include/hz_mlp.h)
        return cfg, InferenceEngine(net, cfg.value_support.max, dtype=torch.float32, device="cuda", fused="fp16x2")
    return cfg, InferenceEngine(net, cfg.value_support.max, dtype=dtype, device="cuda")


def _oracle_search(cfg, eng, tree, hidden0, sims):
    """run_multi replayed through the oracle tree (tests/oracle_replay.py): the GEMM chain per simulation for fp32 engines, the
    stand-alone fused MFMA inference for the 16-bit ones."""
    from tests.oracle_replay import oracle_search
    oracle_search(cfg, eng, tree, hidden0, sims)


# fp32: the launch-per-phase search; fp16: the benched engine -- root inference through the fused tail, ONE persistent search kernel;
# fp16x2: launch-per-phase search around the hand-written recurrent inference that stays within 1e-3 of the reference's fp32 nets
ENGINES = [("Hanabi-Small", torch.float32), ("Hanabi-Small", torch.float16), ("Hanabi-Full", torch.float16), ("Hanabi-Full", "fp16x2")]


@pytest.mark.parametrize("game,dtype", ENGINES)
def test_evaluation_loop_matches_oracle_replay(game, dtype):
    """core/test.py protocol: no noise, deterministic actions -> fully reproducible; every game played to its end."""
    from hanabizero_amd.evaluate import test as run_test
    from oracle.cport import OracleEnv, OracleTree
    cfg, eng = _engine(game, 10 if game == "Hanabi-Small" else 16, 2 if game == "Hanabi-Small" else 4, dtype)
    assert (eng.fused is not None) == (dtype is not torch.float32)
    E, A, S, stack = 12 if game == "Hanabi-Small" else 21, cfg.action_space_size, cfg.num_simulations, cfg.stacked_observations
    scores, steps = run_test(cfg, eng, test_episodes=E, tie_seed=5)
    env = OracleEnv(game, np.arange(E))
    env.reset()
    obs, legal = env.observe()
    windows = [[obs[i].copy() for _ in range(stack)] for i in range(E)]
    done = np.zeros(E, bool)
    final, nsteps = np.zeros(E, np.int64), np.zeros(E, np.int64)
    D = cfg.obs_dim
    Dp = eng.pad_observations(D, stack)  # (the actor's window layout: slots padded to 16-B multiples, zero weights on the pad)
    while not done.all():
        win = np.zeros((E, stack, Dp), np.float32)
        win[:, :, :D] = np.stack([np.concatenate(w) for w in windows]).reshape(E, stack, D)
        _, l0, h0 = eng.initial(torch.from_numpy(win.reshape(E, -1)).cuda(), padded=Dp != D)
        tree = OracleTree(E, A, S, seed=5, value_delta_max=cfg.value_delta_max)
        tree.prepare_no_noise(np.zeros(E, np.float32), l0.cpu().numpy(), legal)
        _oracle_search(cfg, eng, tree, h0, S)
        dist = tree.distributions() * (legal != 0)
        act = dist.argmax(1).astype(np.int32)
        active = ~done
        rew, d, sc = env.step(act, active.astype(np.uint8))
        obs, legal = env.observe()
        for i in np.nonzero(active)[0]:
            nsteps[i] += 1
            windows[i] = windows[i][1:] + [obs[i].copy()]
            if d[i]:
                done[i], final[i] = True, sc[i]
    assert scores == final.tolist() and steps == nsteps.tolist()


@pytest.mark.parametrize("game,dtype", ENGINES)
def test_reanalyze_policy_targets_match_oracle_replay(game, dtype):
    from hanabizero_amd.reanalyze import prepare_policy_re
    from oracle.cport import OracleTree
    cfg, eng = _engine(game, 12 if game == "Hanabi-Small" else 50, 2 if game == "Hanabi-Small" else 4, dtype)
    A, U = cfg.action_space_size, cfg.num_unroll_steps + 1
    P = 5
    B = P * U
    rng = np.random.RandomState(0)
    obs = (rng.rand(B, cfg.obs_shape) < 0.3).astype(np.float32)
    legal = (rng.rand(B, A) < 0.6).astype(np.float64)
    legal[:, 0] = 1
    mask = (rng.rand(B) < 0.8).astype(np.int64)
    noises = rng.dirichlet([0.3] * A, B).astype(np.float32)
    ctx = (obs, mask, list(range(P)), list(range(P)), None, None, [l for l in legal])
    got = prepare_policy_re(cfg, eng, ctx, noises=noises, tie_seed=9)
    assert got.shape == (P, U, A)
    _, l0, h0 = eng.initial(torch.from_numpy(obs).cuda())
    tree = OracleTree(B, A, cfg.num_simulations, seed=9, value_delta_max=cfg.value_delta_max)
    tree.prepare(cfg.root_exploration_fraction, noises * legal.astype(np.float32), np.zeros(B, np.float32), l0.cpu().numpy(),
                 legal.astype(np.int32))
    _oracle_search(cfg, eng, tree, h0, cfg.num_simulations)
    d = tree.distributions().astype(np.float64)
    want = np.where(mask[:, None] != 0, d / d.sum(1, keepdims=True), 0.0).reshape(P, U, A)
    assert np.array_equal(got, want)


def test_callers_run_on_the_fused_bf16_engine():
    """the MFMA-kernel engine drives the same callers (no oracle parity here: bf16 nets; scores are sane and final)"""
    from hanabizero_amd.evaluate import test as run_test
    from hanabizero_amd.reanalyze import prepare_policy_re
    cfg, eng = _engine("Hanabi-Full", 12, 4, dtype=torch.bfloat16)
    assert eng.fused is not None
    scores, steps = run_test(cfg, eng, test_episodes=48, tie_seed=1)
    assert len(scores) == 48 and all(0 <= s <= 25 for s in scores) and all(1 <= n <= cfg.max_moves for n in steps)
    A, U = cfg.action_space_size, cfg.num_unroll_steps + 1
    B = 4 * U
    rng = np.random.RandomState(1)
    ctx = ((rng.rand(B, cfg.obs_shape) < 0.2).astype(np.float32), np.ones(B, np.int64), list(range(4)), list(range(4)), None, None,
           [np.ones(A) for _ in range(B)])
    pol = prepare_policy_re(cfg, eng, ctx, tie_seed=2)
    assert pol.shape == (4, U, A) and np.allclose(pol.sum(-1), 1.0)
