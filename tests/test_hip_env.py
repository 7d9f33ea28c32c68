"""GPU: the HIP Hanabi env, called through the C ABI, against the golden streams of the compiled reference, the
plain-C oracle on long random play (incl. RNG block regeneration), and properties at BASELINE.json's full size."""
import numpy as np
import pytest
import torch

from tests.scenarios import env_fixtures, load_env, replay_env_streams

pytestmark = pytest.mark.gpu


def make_hip(game, seeds):
    from tests.hip_adapters import HipEnv
    return HipEnv(game, seeds)


@pytest.mark.parametrize("game", env_fixtures())
def test_hip_env_matches_reference_golden(game):
    n, T = replay_env_streams(make_hip, game, load_env(game))
    assert n >= 16


def _random_play(game, N, steps, seed):
    """HIP vs oracle, N envs, uniformly random legal play with resets; long enough that every env regenerates its
    mt19937 block (624 draws = 312 deals) at least once when steps is large."""
    from oracle.cport import OracleEnv
    seeds = np.arange(N) + seed * 1000
    H, O = make_hip(game, seeds), OracleEnv(game, seeds)
    H.reset(), O.reset()
    rng = np.random.RandomState(seed)
    done = np.zeros(N, np.uint8)
    for t in range(steps):
        if done.any():
            H.reset(done), O.reset(done)
        ho, hl = H.observe()
        oo, ol = O.observe()
        assert (hl == ol).all(), ("legal", t)
        bad = np.nonzero((ho != oo).any(1))[0]
        assert bad.size == 0, ("obs", t, bad[:5], np.nonzero(ho[bad[0]] != oo[bad[0]])[0][:10])
        assert (H.probe() == O.probe()).all(), ("probe", t)
        u = rng.rand(N, ol.shape[1]) * ol
        act = u.argmax(1).astype(np.int32)
        hr, hd, hs = H.step(act)
        orr, od, osc = O.step(act)
        assert (hr == orr).all() and (hd == od).all() and (hs == osc).all(), ("step", t)
        done = od
    return H


@pytest.mark.parametrize("game,N,steps", [("Hanabi-Small", 256, 120), ("Hanabi-Full", 256, 150), ("Hanabi-Full-5p", 128, 150)])
def test_hip_env_matches_oracle_random_play(game, N, steps):
    _random_play(game, N, steps, 1)


def test_hip_env_long_run_crosses_rng_regeneration():
    # Small deals ~1 card per step; 4 envs x 1500 steps >> 312 deals each
    _random_play("Hanabi-Small", 4, 1500, 2)


def test_observe_dtypes_strides_packed_and_local():
    from hanabizero_amd.hanabi_env import HanabiVecEnv
    N = 300
    g = HanabiVecEnv("Hanabi-Full", np.arange(N))
    loc = HanabiVecEnv("Hanabi-Full", np.arange(N), mdp="local")
    g.reset(), loc.reset()
    ref, legal = g.observe()
    assert ref.shape == (N, 785) and legal.shape == (N, 20) and ref.dtype == torch.uint8
    lobs, _ = loc.observe()
    assert lobs.shape == (N, 660) and torch.equal(lobs, ref[:, 125:])
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        buf = torch.full((N, 4, 800), 7, dtype=dt, device="cuda")  # a stacked-observation ring: write slot 2
        g.observe(out=buf[:, 2, :])
        assert torch.equal(buf[:, 2, :785].to(torch.uint8), ref)
        assert (buf[:, 2, 785:] == 7).all() and (buf[:, 1] == 7).all() and (buf[:, 3] == 7).all()
    packed = torch.zeros((N, g.packed_words), dtype=torch.int32, device="cuda")
    g.observe_packed(packed, legal)
    bits = np.unpackbits(packed.cpu().numpy().view(np.uint8), axis=1, bitorder="little")[:, :785]
    assert (bits == ref.cpu().numpy()).all()


def test_full_size_properties():
    """4096 Hanabi-Full envs, random legal play: card conservation, score bounds, reward = score delta, and
    seed determinism (env i of a 4096-batch == env 0 of a 1-batch with the same seed)."""
    from hanabizero_amd.hanabi_env import HanabiVecEnv
    N = 4096
    g = HanabiVecEnv("Hanabi-Full", np.arange(N))
    one = HanabiVecEnv("Hanabi-Full", [1234])
    g.reset(), one.reset()
    gen = torch.Generator(device="cuda").manual_seed(0)
    prev_score = torch.zeros(N, dtype=torch.int32, device="cuda")
    for t in range(60):
        obs, legal = g.observe()
        o1, l1 = one.observe()
        assert torch.equal(obs[1234], o1[0]) and torch.equal(legal[1234], l1[0])
        pr = g.probe()
        own = obs[:, :125].view(N, 5, 25).sum((1, 2))
        other = obs[:, 125:250].view(N, 5, 25).sum((1, 2))
        disc = obs[:, 125 + 127 + 76:125 + 127 + 76 + 50].sum(1)
        fw = pr[:, 4:9].sum(1)
        # every card is in the deck, a hand, the discard pile or a firework (lost-life plays go to the discard pile)
        assert ((pr[:, 1] + own + other + disc + fw) == 50).all(), t
        assert (pr[:, 15] <= 25).all() and (pr[:, 3] <= 3).all() and (pr[:, 2] <= 8).all()
        assert (legal.sum(1) > 0).all()
        act = (torch.rand(legal.shape, device="cuda", generator=gen) * legal).argmax(1).int()
        reward, done, score, status = g.step(act)
        one.step(act[1234:1235])
        assert (status == 0).all()
        assert torch.equal(reward, score - prev_score)
        prev_score = score.clone()
        if done.any():
            g.reset(done)
            prev_score[done.bool()] = 0
        if one.done.any():
            one.reset()


def test_illegal_move_sets_status_and_leaves_state():
    from hanabizero_amd.hanabi_env import HanabiVecEnv
    g = HanabiVecEnv("Hanabi-Full", [0, 1])
    g.reset()
    before = g.probe().clone()
    obs0, _ = g.observe()
    _, _, _, status = g.step(torch.tensor([0, 5], dtype=torch.int32))  # discard at 8 info tokens: illegal; play 0: legal
    assert status.tolist() == [1, 0]
    after = g.probe()
    assert torch.equal(before[0], after[0]) and not torch.equal(before[1], after[1])
    assert torch.equal(g.observe()[0][0], obs0[0])


def test_reference_python_api_wrapper():
    """config.new_game(seed).reset()/step() shapes and types (env_wrapper.py:18-31) against the golden stream."""
    from hanabizero_amd.hanabi_env import HanabiControlWrapper, HanabiEnv
    from tests.scenarios import env_streams
    fx = load_env("Hanabi-Small")
    key, seed, s = next(x for x in env_streams(fx) if x[0] == "s7_smart")
    env = HanabiControlWrapper(HanabiEnv({"hanabi_name": "Hanabi-Small", "seed": seed}), discount=0.999, mdp="global")
    assert env.action_space_size == 11 and env.env.action_space.n == 11
    for t in range(len(s["action"])):
        if s["action"][t] < 0:
            obs, legal = env.reset()
        else:
            obs, reward, done, info, legal = env.step(int(s["action"][t]))
            assert reward.shape == () and int(reward) == s["reward"][t] and bool(done) == bool(s["done"][t])
            assert info.item()["score"] == s["score"][t]
        assert obs.shape == (193,) and (obs == s["obs"][t]).all() and (legal == s["legal"][t]).all()
