"""tests/scenarios.py -- scenario runners shared by the oracle tests (CPU) and the HIP parity tests (GPU).

A "tree implementation" is any object with the method set of oracle.cport.OracleTree / oracle.ref.RefTree /
tests.hip_adapters.HipTree; an "env implementation" likewise follows oracle.cport.OracleEnv.
"""
import glob
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tree_fixtures():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLD, "tree_*.npz")))


def load_tree(name):
    return dict(np.load(os.path.join(GOLD, "tree_%s.npz" % name), allow_pickle=False))


def bits(a):
    a = np.ascontiguousarray(a, np.float32)
    return a.view(np.uint32)


def run_tree_fixture(make_tree, fx, check_each_sim=True):
    """Drive one golden scenario through `make_tree(N, A, S, seed, delta)`; assert bit-exact agreement."""
    N, A, S = int(fx["N"]), int(fx["A"]), int(fx["S"])
    t = make_tree(N, A, S, int(fx["tie_seed"]), float(fx["value_delta_max"]))
    if int(fx["with_noise"]):
        t.prepare(float(fx["frac"]), fx["noises"], fx["root_rewards"], fx["root_logits"], fx["legal"])
    else:
        t.prepare_no_noise(fx["root_rewards"], fx["root_logits"], fx["legal"])
    assert (bits(t.root_priors()) == bits(fx["out_root_priors"])).all(), "root priors"
    base, init, disc = int(fx["pb_c_base"]), float(fx["pb_c_init"]), float(fx["discount"])
    for sim in range(S - 1):
        ix, iy, la = t.traverse(sim, base, init, disc)
        if check_each_sim:
            assert (ix == fx["out_ix"][sim]).all(), ("ix", sim)
            assert (iy == fx["out_iy"][sim]).all(), ("iy", sim)
            assert (la == fx["out_last_action"][sim]).all(), ("last_action", sim)
            assert (t.path_len() == fx["out_path_len"][sim]).all(), ("path_len", sim)
        t.backprop(sim + 1, disc, fx["rewards"][sim], fx["values"][sim], fx["logits"][sim])
        if check_each_sim:
            mn, mx = t.minmax()
            assert (mn == fx["out_min"][sim]).all(), ("min", sim)
            assert (mx == fx["out_max"][sim]).all(), ("max", sim)
    assert (t.distributions() == fx["out_distributions"]).all(), "distributions"
    assert (bits(t.values()) == bits(fx["out_values"])).all(), "values"
    assert (t.trajectories(S) == fx["out_trajectories"]).all(), "trajectories"
    return t


def env_fixtures():
    return ["Hanabi-Small", "Hanabi-Full", "Hanabi-Full-5p"]


def load_env(game):
    return dict(np.load(os.path.join(GOLD, "env_%s.npz" % game), allow_pickle=False))


def env_streams(fx):
    """yields (key, seed, dict of arrays) per recorded (seed, policy) stream."""
    D = int(fx["own_len"]) + int(fx["obs_len"]) + int(fx["players"])
    A = int(fx["num_moves"])
    for key in fx["keys"]:
        key = str(key)
        seed = int(key.split("_")[0][1:])
        s = {k: fx[key + "_" + k] for k in ("action", "reward", "done", "score", "probe")}
        s["legal"] = np.unpackbits(fx[key + "_legal"], axis=1)[:, :A]
        s["obs"] = np.unpackbits(fx[key + "_obs"], axis=1)[:, :D]
        yield key, seed, s


def replay_env_streams(make_env, game, fx, keys=None):
    """Replay every recorded stream of a game in ONE batched env (stream i = env i); assert bit-exact agreement.

    Streams have different lengths: finished streams are masked out.  Rows with action -1 are resets.
    """
    streams = [(k, seed, s) for k, seed, s in env_streams(fx) if keys is None or k in keys]
    env = make_env(game, [seed for _, seed, _ in streams])
    n = len(streams)
    T = max(len(s["action"]) for _, _, s in streams)
    for t in range(T):
        alive = np.array([t < len(s["action"]) for _, _, s in streams])
        act = np.array([s["action"][t] if a else 0 for a, (_, _, s) in zip(alive, streams)], np.int32)
        is_reset = alive & (act < 0)
        is_step = alive & (act >= 0)
        if is_reset.any():
            env.reset(is_reset.astype(np.uint8))
        if is_step.any():
            reward, done, score = env.step(np.where(is_step, act, 0), is_step.astype(np.uint8))
        obs, legal = env.observe()
        probe = env.probe()
        for i, (key, _, s) in enumerate(streams):
            if not alive[i]:
                continue
            ctx = (game, key, t)
            if is_step[i]:
                assert reward[i] == s["reward"][t], ("reward",) + ctx
                assert bool(done[i]) == bool(s["done"][t]), ("done",) + ctx
                assert score[i] == s["score"][t], ("score",) + ctx
            assert (probe[i] == s["probe"][t]).all(), ("probe",) + ctx + (probe[i], s["probe"][t])
            assert (legal[i] == s["legal"][t]).all(), ("legal",) + ctx
            bad = np.nonzero(obs[i] != s["obs"][t])[0]
            assert bad.size == 0, ("obs",) + ctx + (bad[:10],)
    return n, T
