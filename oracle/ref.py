"""oracle/ref.py -- TEST INFRASTRUCTURE: ctypes drivers for the GENUINE reference builds in oracle/_ref/.

* ``RefTree``      drives oracle/_ref/libref_tree.so (reference core/ctree/cnode.cpp + cminimax.cpp behind
                   oracle/ref_tree_harness.cpp) with the call sequence of core/ctree/cytree.pyx.
* ``RefHanabiEnv`` drives oracle/_ref/libpyhanabi.so (reference envs/hanabi/pyhanabi.cc + hanabi_lib/*.cc) through
                   the same C API envs/hanabi/pyhanabi.py binds with cffi, and restates the composition done in
                   envs/hanabi/rl_env.py:148-267 (reset) and :292-442 (step): share_obs = own_hand ++ canonical ++
                   onehot(cur_player), obs = canonical ++ onehot(cur_player), legal mask, reward = score delta.

Only tests/, tools/gen_golden.py and bench.py's cpu_baseline leg may import this module.  The product
(hanabizero_amd/) never does.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.join(_HERE, "_ref")


def ref_available():
    return (os.path.exists(os.path.join(_REF_DIR, "libref_tree.so"))
            and os.path.exists(os.path.join(_REF_DIR, "libpyhanabi.so")))


_tree_lib = None
_hanabi_lib = None


def _tree():
    global _tree_lib
    if _tree_lib is None:
        lib = C.CDLL(os.path.join(_REF_DIR, "libref_tree.so"))
        lib.ref_tree_new.restype = C.c_void_p
        lib.ref_tree_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64]
        lib.ref_tree_free.argtypes = [C.c_void_p]
        lib.ref_tree_set_delta.argtypes = [C.c_void_p, C.c_float]
        lib.ref_tree_prepare.argtypes = [C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 4
        lib.ref_tree_traverse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float] + [C.c_void_p] * 3
        lib.ref_tree_path_len.argtypes = [C.c_void_p, C.c_void_p]
        lib.ref_tree_backprop.argtypes = [C.c_void_p, C.c_int, C.c_float] + [C.c_void_p] * 3
        lib.ref_tree_distributions.argtypes = [C.c_void_p, C.c_void_p]
        lib.ref_tree_values.argtypes = [C.c_void_p, C.c_void_p]
        lib.ref_tree_trajectories.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_tree_minmax.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_tree_root_priors.argtypes = [C.c_void_p, C.c_void_p]
        _tree_lib = lib
    return _tree_lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefTree:
    """cytree.Roots + MinMaxStatsList + ResultsWrapper of the reference, array-in/array-out."""

    def __init__(self, N, A, S, mode=1, seed=0, value_delta_max=0.006):
        self.lib = _tree()
        self.N, self.A, self.S = N, A, S
        self.h = self.lib.ref_tree_new(N, A, S, mode, seed)
        self.lib.ref_tree_set_delta(self.h, value_delta_max)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_tree_free(self.h)
            self.h = None

    def prepare(self, frac, noises, rewards, logits, legal):
        n = np.ascontiguousarray(noises, np.float32)
        r = np.ascontiguousarray(rewards, np.float32)
        l = np.ascontiguousarray(logits, np.float32)
        g = np.ascontiguousarray(legal, np.int32)
        self.lib.ref_tree_prepare(self.h, 1, frac, _p(n), _p(r), _p(l), _p(g))

    def prepare_no_noise(self, rewards, logits, legal):
        r = np.ascontiguousarray(rewards, np.float32)
        l = np.ascontiguousarray(logits, np.float32)
        g = np.ascontiguousarray(legal, np.int32)
        self.lib.ref_tree_prepare(self.h, 0, 0.0, None, _p(r), _p(l), _p(g))

    def traverse(self, sim, pb_c_base, pb_c_init, discount):
        ix = np.empty(self.N, np.int32)
        iy = np.empty(self.N, np.int32)
        la = np.empty(self.N, np.int32)
        self.lib.ref_tree_traverse(self.h, sim, pb_c_base, pb_c_init, discount, _p(ix), _p(iy), _p(la))
        return ix, iy, la

    def path_len(self):
        out = np.empty(self.N, np.int32)
        self.lib.ref_tree_path_len(self.h, _p(out))
        return out

    def backprop(self, hidden_state_index_x, discount, rewards, values, logits):
        r = np.ascontiguousarray(rewards, np.float32)
        v = np.ascontiguousarray(values, np.float32)
        l = np.ascontiguousarray(logits, np.float32)
        self.lib.ref_tree_backprop(self.h, hidden_state_index_x, discount, _p(r), _p(v), _p(l))

    def distributions(self):
        out = np.empty((self.N, self.A), np.int32)
        self.lib.ref_tree_distributions(self.h, _p(out))
        return out

    def values(self):
        out = np.empty(self.N, np.float32)
        self.lib.ref_tree_values(self.h, _p(out))
        return out

    def trajectories(self, max_len=None):
        max_len = max_len or self.S
        out = np.empty((self.N, max_len), np.int32)
        self.lib.ref_tree_trajectories(self.h, _p(out), max_len)
        return out

    def minmax(self):
        mn = np.empty(self.N, np.float32)
        mx = np.empty(self.N, np.float32)
        self.lib.ref_tree_minmax(self.h, _p(mn), _p(mx))
        return mn, mx

    def root_priors(self):
        out = np.empty((self.N, self.A), np.float32)
        self.lib.ref_tree_root_priors(self.h, _p(out))
        return out


# ----------------------------------------------------------------------------------------------
class _Handle(C.Structure):
    _fields_ = [("p", C.c_void_p)]


def _hanabi():
    global _hanabi_lib
    if _hanabi_lib is None:
        lib = C.CDLL(os.path.join(_REF_DIR, "libpyhanabi.so"))
        H = C.POINTER(_Handle)
        lib.NewGame.argtypes = [H, C.c_int, C.POINTER(C.c_char_p)]
        lib.DeleteGame.argtypes = [H]
        lib.NewState.argtypes = [H, H]
        lib.DeleteState.argtypes = [H]
        for name in ("StateCurPlayer", "StateDeckSize", "StateEndOfGameStatus", "StateInformationTokens",
                     "StateLifeTokens", "StateScore", "StateNumPlayers"):
            getattr(lib, name).argtypes = [H]
            getattr(lib, name).restype = C.c_int
        lib.StateFireworks.argtypes = [H, C.c_int]
        lib.StateFireworks.restype = C.c_int
        lib.StateDealRandomCard.argtypes = [H]
        lib.StateApplyMove.argtypes = [H, H]
        lib.GetMoveByUid.argtypes = [H, C.c_int, H]
        lib.DeleteMove.argtypes = [H]
        lib.GetMoveUid.argtypes = [H, H]
        lib.GetMoveUid.restype = C.c_int
        for name in ("MaxMoves", "NumPlayers", "NumColors", "NumRanks", "HandSize"):
            getattr(lib, name).argtypes = [H]
            getattr(lib, name).restype = C.c_int
        lib.NewObservation.argtypes = [H, C.c_int, H]
        lib.DeleteObservation.argtypes = [H]
        lib.ObsNumLegalMoves.argtypes = [H]
        lib.ObsNumLegalMoves.restype = C.c_int
        lib.ObsGetLegalMove.argtypes = [H, C.c_int, H]
        lib.NewObservationEncoder.argtypes = [H, H, C.c_int]
        lib.DeleteObservationEncoder.argtypes = [H]
        for name in ("ObservationShape", "OwnHandShape"):
            getattr(lib, name).argtypes = [H]
            getattr(lib, name).restype = C.c_void_p
        for name in ("EncodeObservation", "EncodeOwnHandObservation"):
            getattr(lib, name).argtypes = [H, H]
            getattr(lib, name).restype = C.c_void_p
        lib.DeleteString.argtypes = [C.c_void_p]
        lib.StateGetHandSize.argtypes = [H, C.c_int]
        lib.StateGetHandSize.restype = C.c_int
        _hanabi_lib = lib
    return _hanabi_lib


GAME_PARAMS = {
    # envs/hanabi/rl_env.py:110-119
    "Hanabi-Full": dict(colors=5, ranks=5, players=2, max_information_tokens=8, max_life_tokens=3,
                        observation_type=1),
    # envs/hanabi/rl_env.py:121-131
    "Hanabi-Small": dict(colors=2, ranks=5, players=2, hand_size=2, max_information_tokens=3,
                         max_life_tokens=1, observation_type=1),
    # BASELINE.json config 5: same C++ library, players=5 (hand size 4 from HandSizeFromRules, hanabi_game.cc:147-152)
    "Hanabi-Full-5p": dict(colors=5, ranks=5, players=5, max_information_tokens=8, max_life_tokens=3,
                           observation_type=1),
}


class RefHanabiEnv:
    """envs/hanabi/rl_env.py HanabiEnv restated over the reference C API (ctypes instead of cffi)."""

    def __init__(self, name, seed):
        self.lib = lib = _hanabi()
        params = dict(GAME_PARAMS[name])
        params["seed"] = 0 if seed is None else seed  # rl_env.py:106-109
        kv = []
        for k, v in params.items():
            kv += [str(k).encode(), str(v).encode()]
        arr = (C.c_char_p * len(kv))(*kv)
        self.game = _Handle()
        lib.NewGame(C.byref(self.game), len(kv), arr)
        self.enc = _Handle()
        lib.NewObservationEncoder(C.byref(self.enc), C.byref(self.game), 0)
        self.players = lib.NumPlayers(C.byref(self.game))
        self.num_moves = lib.MaxMoves(C.byref(self.game))
        self.num_colors = lib.NumColors(C.byref(self.game))
        self.num_ranks = lib.NumRanks(C.byref(self.game))
        self.hand_size = lib.HandSize(C.byref(self.game))
        self.obs_len = int(self._str(lib.ObservationShape(C.byref(self.enc))))
        self.own_len = int(self._str(lib.OwnHandShape(C.byref(self.enc))))
        self.state = None

    def _str(self, ptr):
        s = C.cast(ptr, C.c_char_p).value.decode()
        self.lib.DeleteString(ptr)
        return s

    def close(self):
        if self.state is not None:
            self.lib.DeleteState(C.byref(self.state))
            self.state = None

    def _deal_all(self):
        while self.lib.StateCurPlayer(C.byref(self.state)) == -1:  # CHANCE_PLAYER_ID
            self.lib.StateDealRandomCard(C.byref(self.state))

    def _observe(self):
        lib = self.lib
        cur = lib.StateCurPlayer(C.byref(self.state))
        ob = _Handle()
        lib.NewObservation(C.byref(self.state), cur, C.byref(ob))
        vec = np.array(self._str(lib.EncodeObservation(C.byref(self.enc), C.byref(ob))).split(","), dtype=np.int64)
        own_s = self._str(lib.EncodeOwnHandObservation(C.byref(self.enc), C.byref(ob)))
        own = np.array(own_s.split(","), dtype=np.int64) if own_s else np.zeros(0, np.int64)
        legal = np.zeros(self.num_moves, np.float64)
        for i in range(lib.ObsNumLegalMoves(C.byref(ob))):
            mv = _Handle()
            lib.ObsGetLegalMove(C.byref(ob), i, C.byref(mv))
            legal[lib.GetMoveUid(C.byref(self.game), C.byref(mv))] = 1
            lib.DeleteMove(C.byref(mv))
        lib.DeleteObservation(C.byref(ob))
        turn = np.zeros(self.players, np.int64)
        turn[cur] = 1
        obs = np.concatenate([vec, turn])
        share = np.concatenate([own, vec, turn])
        return share, obs, legal

    def reset(self):
        """rl_env.py:148-267 -> (share_obs, obs, legal)"""
        self.close()
        self.state = _Handle()
        self.lib.NewState(C.byref(self.game), C.byref(self.state))
        self._deal_all()
        return self._observe()

    def step(self, action):
        """rl_env.py:292-442 -> (share_obs, obs, reward, done, score, legal)"""
        lib = self.lib
        mv = _Handle()
        lib.GetMoveByUid(C.byref(self.game), int(action), C.byref(mv))
        last = lib.StateScore(C.byref(self.state))
        lib.StateApplyMove(C.byref(self.state), C.byref(mv))
        lib.DeleteMove(C.byref(mv))
        self._deal_all()
        share, obs, legal = self._observe()
        done = lib.StateEndOfGameStatus(C.byref(self.state)) != 0
        score = lib.StateScore(C.byref(self.state))
        return share, obs, score - last, done, score, legal

    # state probes used by the golden fixtures
    def probe(self):
        lib, s = self.lib, C.byref(self.state)
        return dict(cur_player=lib.StateCurPlayer(s), deck_size=lib.StateDeckSize(s),
                    info=lib.StateInformationTokens(s), life=lib.StateLifeTokens(s),
                    fireworks=[lib.StateFireworks(s, c) for c in range(self.num_colors)],
                    hand_sizes=[lib.StateGetHandSize(s, p) for p in range(self.players)],
                    status=lib.StateEndOfGameStatus(s), score=lib.StateScore(s))
