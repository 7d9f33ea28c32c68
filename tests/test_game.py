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


def _same_history(a, b):
    for k in ("obs_history", "actions", "rewards", "child_visits", "root_values", "legal_actions"):
        x, y = np.asarray(getattr(a, k)), np.asarray(getattr(b, k))
        assert x.shape == y.shape and np.array_equal(x, y), k
    assert np.asarray(a.child_visits).dtype == np.float64 and np.asarray(a.root_values).dtype == np.float64
    assert len(a) == len(b)


def test_from_packed_equals_the_move_by_move_build_and_ingest_equals_save_game():
    """The vectorised replay ingest (GameHistory.from_packed / from_arrays, ReplayBuffer.ingest_packed) leaves exactly
    what the reference's move-by-move construction (init, store_search_stats + append per move, game_over), put()'s reward
    reshape and save_game leave (core/game.py:170-200, selfplay_worker.py:32-37, replay_buffer.py:108-132)."""
    from hanabizero_amd.replay import ReplayBuffer
    from hanabizero_amd.selfplay import pack_records, unpack_packed, unpack_record
    cfg = make_config("Hanabi-Small", stack=3)
    A, D = cfg.action_space_size, cfg.obs_dim
    W = (D + 31) // 32
    rng = np.random.RandomState(4)
    n, T = 37, 12
    lens = rng.randint(1, T + 1, n)
    rec = dict(action=rng.randint(0, A, (n, T)).astype(np.int8), reward=rng.randint(-2, 3, (n, T)).astype(np.int8),
               value=rng.randn(n, T).astype(np.float32), visits=rng.randint(0, 9, (n, T, A)).astype(np.int16),
               legal=rng.randint(0, 2, (n, T + 1, A)).astype(np.uint8),
               obs=rng.randint(-2**31, 2**31 - 1, (n, T + 1, W)).astype(np.int32),
               meta=np.stack([lens, rng.randint(0, 11, n), np.arange(n), np.zeros(n, np.int64)], 1).astype(np.int32))
    rec["visits"][:, :, 0] += 1
    buf, n_, moves = pack_records(rec, A, W)
    view = unpack_packed(buf, n_, moves, A, W)
    want = ReplayBuffer(cfg)
    want.save_game(GameHistory.from_packed_stepwise(unpack_record(view, 0), None, cfg), True, 0, None)  # (an older game)
    got = ReplayBuffer(cfg)
    got.save_game(GameHistory.from_packed_stepwise(unpack_record(view, 0), None, cfg), True, 0, None)
    for i in range(n):
        r = unpack_record(view, i)
        slow = GameHistory.from_packed_stepwise(r, None, cfg)
        _same_history(GameHistory.from_packed(r, None, cfg), slow)
        want.save_game(reshape_turn_rewards(slow), True, 0, None)
    assert got.ingest_packed(buf, n_, moves) == n
    assert got.size() == want.size() and got.episodes_collected() == want.episodes_collected()
    assert np.array_equal(got.priorities, want.priorities) and got.game_look_up == want.game_look_up
    for a, b in zip(got.buffer, want.buffer):
        _same_history(a, b)
    g, pos, idx, w, mt = got.prepare_batch_context(8, beta=0.4)
    assert len(g) == 8 and all(0 <= p < len(x) for x, p in zip(g, pos))
