"""CPU: GameHistory / reward reshape / config mirrors of the reference's host-side interfaces."""
import numpy as np

from hanabizero_amd.config import make_config
from hanabizero_amd.game import GameHistory, reshape_turn_rewards


def test_config_shapes_and_constants():
    for name, A, D, sup in [("Hanabi-Small", 11, 193, 25), ("Hanabi-Full", 20, 785, 100), ("Hanabi-Full-5p", 48, 1385, 100)]:
        cfg = make_config(name)
        assert (cfg.action_space_size, cfg.obs_dim, cfg.obs_shape) == (A, D, D * 4)
        assert (cfg.pb_c_base, cfg.pb_c_init, cfg.root_exploration_fraction) == (19652, 1.25, 0.25)  # core/config.py:107-111
        assert (cfg.discount, cfg.value_delta_max, cfg.root_dirichlet_alpha) == (0.999, 0.006, 0.3)
        assert cfg.value_support.size == 2 * sup + 1
    local = make_config("Hanabi-Full", mdp_type="local")
    assert local.obs_dim == 660


def test_reward_reshape_matches_put():
    """selfplay_worker.py:32-37: r'[t] = r[t] + r[t-1] using the ORIGINAL previous reward."""
    class G:
        pass
    g = G()
    g.rewards = np.array([1, 0, 2, -3, 0])
    reshape_turn_rewards(g)
    assert g.rewards.tolist() == [1, 1, 2, -1, -3]


def test_from_packed_round_trip():
    cfg = make_config("Hanabi-Small", stack=2)
    T, A, D = 5, cfg.action_space_size, cfg.obs_dim
    rng = np.random.RandomState(0)
    frames = rng.randint(0, 2, (T + 1, D)).astype(np.uint8)
    W = (D + 31) // 32
    bits = np.zeros((T + 1, W * 32), np.uint8)
    bits[:, :D] = frames
    obs_bits = np.packbits(bits, axis=1, bitorder="little").view(np.uint32)
    visits = rng.randint(0, 5, (T, A)).astype(np.int16)
    visits[:, 0] += 1
    rec = dict(len=T, action=rng.randint(0, A, T).astype(np.int8), reward=rng.randint(-1, 2, T).astype(np.int8),
               value=rng.rand(T).astype(np.float32), visits=visits, legal=rng.randint(0, 2, (T + 1, A)).astype(np.uint8),
               obs_bits=obs_bits)
    g = GameHistory.from_packed(rec, None, cfg)
    assert len(g) == T and g.obs_history.shape == (T + 2, D)
    assert (g.obs_history[0] == frames[0]).all() and (g.obs_history[1:] == frames).all()
    assert np.allclose(g.child_visits, visits / visits.sum(1, keepdims=True))
    assert (g.actions == rec["action"]).all() and (g.rewards == rec["reward"]).all()
    assert (np.array(g.obs(1)) == frames[0:2]).all() and len(g.obs(T, extra_len=3, padding=True)) == 5
    assert g.legal_actions.shape == (T + 1, A)
