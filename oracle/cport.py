"""oracle/cport.py -- TEST INFRASTRUCTURE: ctypes drivers for oracle/libhz_oracle.so, the plain-C CPU
restatement (tree_oracle.c, env_oracle.c, mt_discrete.c) of the reference's tree and Hanabi env.

Same method names as oracle/ref.py's RefTree so tests can run one scenario through the genuine
reference, the C restatement and the HIP library and compare arrays.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libhz_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        V = C.c_void_p
        L.hzo_tree_new.restype = V
        L.hzo_tree_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32]
        L.hzo_tree_free.argtypes = [V]
        L.hzo_tree_set_delta.argtypes = [V, C.c_float]
        L.hzo_tree_prepare.argtypes = [V, C.c_int, C.c_float, V, V, V, V]
        L.hzo_tree_traverse.argtypes = [V, C.c_int, C.c_int, C.c_float, C.c_float, V, V, V]
        L.hzo_tree_path_len.argtypes = [V, V]
        L.hzo_tree_backprop.argtypes = [V, C.c_int, C.c_float, V, V, V]
        L.hzo_tree_distributions.argtypes = [V, V]
        L.hzo_tree_values.argtypes = [V, V]
        L.hzo_tree_trajectories.argtypes = [V, V, C.c_int]
        L.hzo_tree_minmax.argtypes = [V, V, V]
        L.hzo_tree_root_priors.argtypes = [V, V]
        L.hzo_env_new.restype = V
        L.hzo_env_new.argtypes = [C.c_int] * 7 + [V]
        L.hzo_env_free.argtypes = [V]
        L.hzo_env_dims.argtypes = [V] + [C.POINTER(C.c_int)] * 4
        L.hzo_env_reset.argtypes = [V, V]
        L.hzo_env_step.restype = C.c_int
        L.hzo_env_step.argtypes = [V, V, V, V, V, V]
        L.hzo_env_observe.argtypes = [V, V, V]
        L.hzo_env_probe.argtypes = [V, V]
        L.hzo_env_hands.argtypes = [V, C.c_int, V]
        L.hzo_expf_checksum_range.argtypes = [V, C.c_int, C.c_int]
        L.hzo_expf_array.argtypes = [V, V, C.c_int64]
        L.hzo_mt_seed.argtypes = [V, C.c_uint32]
        L.hzo_mt_next.restype = C.c_uint32
        L.hzo_mt_next.argtypes = [V]
        L.hzo_canonical53.restype = C.c_double
        L.hzo_canonical53.argtypes = [V]
        L.hzo_discrete.restype = C.c_int
        L.hzo_discrete.argtypes = [V, V, C.c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleTree:
    def __init__(self, N, A, S, seed=0, value_delta_max=0.006, tree_id_base=0):
        self.lib = lib()
        self.N, self.A, self.S = N, A, S
        self.h = self.lib.hzo_tree_new(N, A, S, seed, tree_id_base)
        self.lib.hzo_tree_set_delta(self.h, value_delta_max)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.hzo_tree_free(self.h)
            self.h = None

    def prepare(self, frac, noises, rewards, logits, legal):
        n = np.ascontiguousarray(noises, np.float32)
        r = np.ascontiguousarray(rewards, np.float32)
        l = np.ascontiguousarray(logits, np.float32)
        g = np.ascontiguousarray(legal, np.int32)
        self.lib.hzo_tree_prepare(self.h, 1, frac, _p(n), _p(r), _p(l), _p(g))

    def prepare_no_noise(self, rewards, logits, legal):
        r = np.ascontiguousarray(rewards, np.float32)
        l = np.ascontiguousarray(logits, np.float32)
        g = np.ascontiguousarray(legal, np.int32)
        self.lib.hzo_tree_prepare(self.h, 0, 0.0, None, _p(r), _p(l), _p(g))

    def traverse(self, sim, pb_c_base, pb_c_init, discount):
        ix = np.empty(self.N, np.int32)
        iy = np.empty(self.N, np.int32)
        la = np.empty(self.N, np.int32)
        self.lib.hzo_tree_traverse(self.h, sim, pb_c_base, pb_c_init, discount, _p(ix), _p(iy), _p(la))
        return ix, iy, la

    def path_len(self):
        out = np.empty(self.N, np.int32)
        self.lib.hzo_tree_path_len(self.h, _p(out))
        return out

    def backprop(self, hidden_state_index_x, discount, rewards, values, logits):
        r = np.ascontiguousarray(rewards, np.float32)
        v = np.ascontiguousarray(values, np.float32)
        l = np.ascontiguousarray(logits, np.float32)
        self.lib.hzo_tree_backprop(self.h, hidden_state_index_x, discount, _p(r), _p(v), _p(l))

    def distributions(self):
        out = np.empty((self.N, self.A), np.int32)
        self.lib.hzo_tree_distributions(self.h, _p(out))
        return out

    def values(self):
        out = np.empty(self.N, np.float32)
        self.lib.hzo_tree_values(self.h, _p(out))
        return out

    def trajectories(self, max_len=None):
        max_len = max_len or self.S
        out = np.empty((self.N, max_len), np.int32)
        self.lib.hzo_tree_trajectories(self.h, _p(out), max_len)
        return out

    def minmax(self):
        mn = np.empty(self.N, np.float32)
        mx = np.empty(self.N, np.float32)
        self.lib.hzo_tree_minmax(self.h, _p(mn), _p(mx))
        return mn, mx

    def root_priors(self):
        out = np.empty((self.N, self.A), np.float32)
        self.lib.hzo_tree_root_priors(self.h, _p(out))
        return out


# reference game parameter sets (envs/hanabi/rl_env.py:110-131; 5p = BASELINE.json config 5)
GAMES = {
    "Hanabi-Small": dict(colors=2, ranks=5, players=2, hand_size=2, max_info=3, max_life=1),
    "Hanabi-Full": dict(colors=5, ranks=5, players=2, hand_size=-1, max_info=8, max_life=3),
    "Hanabi-Full-5p": dict(colors=5, ranks=5, players=5, hand_size=-1, max_info=8, max_life=3),
}


class OracleEnv:
    """N envs of one game; seeds[i] seeds env i's own mt19937 (one HanabiGame per env in the reference)."""

    def __init__(self, name, seeds):
        self.lib = lib()
        g = GAMES[name]
        seeds = np.ascontiguousarray(seeds, np.int32)
        self.N = len(seeds)
        self.h = self.lib.hzo_env_new(self.N, g["colors"], g["ranks"], g["players"], g["hand_size"],
                                      g["max_info"], g["max_life"], _p(seeds))
        a, o, w, p = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.lib.hzo_env_dims(self.h, C.byref(a), C.byref(o), C.byref(w), C.byref(p))
        self.num_moves, self.obs_len, self.own_len, self.players = a.value, o.value, w.value, p.value
        self.D = self.own_len + self.obs_len + self.players
        self.hand_size = self.own_len // (g["colors"] * g["ranks"])

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.hzo_env_free(self.h)
            self.h = None

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.hzo_env_reset(self.h, _p(m))

    def step(self, actions, mask=None):
        a = np.ascontiguousarray(actions, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        reward = np.zeros(self.N, np.int32)
        done = np.zeros(self.N, np.uint8)
        score = np.zeros(self.N, np.int32)
        rc = self.lib.hzo_env_step(self.h, _p(a), _p(m), _p(reward), _p(done), _p(score))
        if rc != 0:
            raise ValueError("illegal move in env %d" % (-rc - 1))
        return reward, done, score

    def observe(self):
        obs = np.empty((self.N, self.D), np.uint8)
        legal = np.empty((self.N, self.num_moves), np.uint8)
        self.lib.hzo_env_observe(self.h, _p(obs), _p(legal))
        return obs, legal

    def probe(self):
        out = np.empty((self.N, 16), np.int32)
        self.lib.hzo_env_probe(self.h, _p(out))
        return out

    def hands(self, env):
        out = np.empty((self.players, self.hand_size), np.int32)
        self.lib.hzo_env_hands(self.h, env, _p(out))
        return out


def expf_checksums(threads=8):
    """checksum per block of 2^20 bit patterns over all 2^32 floats, host libm expf (ctypes drops the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    L = lib()
    out = np.zeros(4096, np.uint64)
    step = 64
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda b: L.hzo_expf_checksum_range(_p(out), b, min(b + step, 4096)), range(0, 4096, step)))
    return out


def expf_array(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().hzo_expf_array(_p(x), _p(y), x.size)
    return y
