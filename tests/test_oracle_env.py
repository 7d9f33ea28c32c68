"""CPU: the plain-C Hanabi restatement (oracle/env_oracle.c + mt_discrete.c) against the golden streams produced by
the compiled reference envs/hanabi (tools/gen_golden.py), and against the reference directly when present."""
import numpy as np
import pytest

from oracle.cport import OracleEnv
from oracle.ref import RefHanabiEnv, ref_available
from tests.scenarios import env_fixtures, load_env, replay_env_streams


@pytest.mark.parametrize("game", env_fixtures())
def test_oracle_env_matches_golden(game):
    fx = load_env(game)
    o = OracleEnv(game, [0])
    assert (o.num_moves, o.obs_len, o.own_len, o.players) == tuple(int(fx[k]) for k in ("num_moves", "obs_len", "own_len", "players"))
    n, T = replay_env_streams(lambda g, seeds: OracleEnv(g, seeds), game, fx)
    assert n >= 16 and T > 50


def test_shapes_match_survey():
    # Small 2p A=11 D=193, Full 2p A=20 D=785, Full 5p A=48 D=1385
    for game, A, D in [("Hanabi-Small", 11, 193), ("Hanabi-Full", 20, 785), ("Hanabi-Full-5p", 48, 1385)]:
        o = OracleEnv(game, [0])
        assert (o.num_moves, o.D) == (A, D)


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("game", env_fixtures())
def test_oracle_env_matches_reference_random_play(game):
    for seed in (3, 99):
        r, o = RefHanabiEnv(game, seed), OracleEnv(game, [seed])
        rng = np.random.RandomState(seed)
        for ep in range(3):
            s, _, legal = r.reset()
            o.reset()
            done = False
            while True:
                so, lo = o.observe()
                assert (so[0] == s).all() and (lo[0] == legal).all()
                if done:
                    break
                a = rng.choice(np.nonzero(legal)[0])
                s, _, rew, done, score, legal = r.step(a)
                ro, do, sc = o.step([a])
                assert (ro[0], bool(do[0]), sc[0]) == (rew, done, score)


def test_illegal_move_is_an_error_not_an_abort():
    o = OracleEnv("Hanabi-Full", [0])
    o.reset()
    _, legal = o.observe()
    bad = int(np.nonzero(legal[0] == 0)[0][0])  # discard at max info tokens is illegal at the start
    with pytest.raises(ValueError):
        o.step([bad])
