"""hanabizero_amd.hanabi_env -- the Hanabi environment of HanabiZero on the GPU.

Two layers, both over libhanabizero_hip.so (include/hz_env.h); there is no CPU implementation here.

* ``HanabiVecEnv``  N games advanced in lock-step with device-resident inputs/outputs: what the MI355X actor uses.
* ``HanabiEnv`` / ``HanabiControlWrapper``  the reference's per-env Python API
  (/root/reference/envs/hanabi/rl_env.py:87-442, config/hanabi_control/env_wrapper.py:6-34): ``reset() ->
  (obs, legal)``, ``step(a) -> (obs, reward, done, info, legal)`` with numpy values, implemented as a 1-env view of
  the same kernels so that ``config.new_game(seed)`` and core/test.py keep working unchanged.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib

# envs/hanabi/rl_env.py:110-131 (the two games the reference hard-codes) + the 5-player game of BASELINE config 5
GAMES = {
    "Hanabi-Full": dict(colors=5, ranks=5, players=2, hand_size=-1, max_information_tokens=8, max_life_tokens=3),
    "Hanabi-Small": dict(colors=2, ranks=5, players=2, hand_size=2, max_information_tokens=3, max_life_tokens=1),
    "Hanabi-Full-5p": dict(colors=5, ranks=5, players=5, hand_size=-1, max_information_tokens=8, max_life_tokens=3),
}
_OBS_DTYPES = {torch.uint8: 0, torch.float32: 1, torch.bfloat16: 2, torch.float16: 3}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class HanabiVecEnv:
    def __init__(self, hanabi_name, seeds, device=None, mdp="global", **overrides):
        if hanabi_name not in GAMES:
            raise ValueError("Unknown environment {}".format(hanabi_name))  # rl_env.py:133
        g = dict(GAMES[hanabi_name])
        g.update(overrides)
        self.game = g
        self.name = hanabi_name
        self.mdp = {"global": 0, "local": 1}[mdp]
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.int64).astype(np.int32))
        self.N = int(seeds.size)
        h = C.c_void_p()
        check(lib.hz_env_create(C.byref(h), self.N, g["colors"], g["ranks"], g["players"], g["hand_size"],
                                g["max_information_tokens"], g["max_life_tokens"],
                                seeds.ctypes.data_as(C.c_void_p), idx), "hz_env_create")
        self._h = h
        a, o, w, p = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib.hz_env_dims(self._h, C.byref(a), C.byref(o), C.byref(w), C.byref(p)), "hz_env_dims")
        self.num_moves, self.obs_len, self.own_len, self.players = a.value, o.value, w.value, p.value
        self.obs_dim = (self.own_len if self.mdp == 0 else 0) + self.obs_len + self.players
        self.packed_words = (self.obs_dim + 31) // 32
        d, N = self.device, self.N
        self.reward = torch.zeros(N, dtype=torch.int32, device=d)
        self.done = torch.zeros(N, dtype=torch.uint8, device=d)
        self.score = torch.zeros(N, dtype=torch.int32, device=d)
        self.status = torch.zeros(N, dtype=torch.int32, device=d)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and lib is not None:
            lib.hz_env_destroy(h)
            self._h = None

    @property
    def hbm_bytes(self):
        return int(lib.hz_env_hbm_bytes(self._h))

    def reset(self, mask=None, rows=None):
        """mask: None (all) or uint8/bool CUDA tensor [N].  rows: optional _lib.RowsJob (include/hz_rows.h), an independent
        row scatter the same launch carries along (hz_env_reset_rows)."""
        m = None if mask is None else self._mask(mask)
        if rows is not None:
            import ctypes as C
            check(lib.hz_env_reset_rows(self._h, None if m is None else m.data_ptr(), C.byref(rows), _stream()), "hz_env_reset_rows")
        else:
            check(lib.hz_env_reset(self._h, None if m is None else m.data_ptr(), _stream()), "hz_env_reset")

    def _mask(self, mask):
        m = mask if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask), device=self.device)
        m = m.to(self.device).to(torch.uint8)
        return m if m.is_contiguous() else m.contiguous()

    def step(self, actions, mask=None):
        """actions: int CUDA tensor [N] of move uids.  Returns (reward i32, done u8, score i32, status i32) device
        tensors (buffers owned by the env, overwritten by the next step)."""
        a = actions if isinstance(actions, torch.Tensor) else torch.as_tensor(np.asarray(actions), device=self.device)
        a = a.to(self.device).to(torch.int32).contiguous()
        m = None if mask is None else self._mask(mask)
        check(lib.hz_env_step(self._h, a.data_ptr(), None if m is None else m.data_ptr(), self.reward.data_ptr(),
                              self.done.data_ptr(), self.score.data_ptr(), self.status.data_ptr(), _stream()),
              "hz_env_step")
        self._keep = (a, m)
        return self.reward, self.done, self.score, self.status

    def observe(self, out=None, packed=None, legal=None, dtype=torch.uint8):
        """Encode the current player's observation of every env.
        out: [N, >=obs_dim] tensor (row stride arbitrary, element stride 1) or None to allocate [N, obs_dim] of dtype.
        packed: optional int32 [N, packed_words]; legal: optional uint8 [N, num_moves].  Returns (out, legal)."""
        if out is None:
            out = torch.empty((self.N, self.obs_dim), dtype=dtype, device=self.device)
        if legal is None:
            legal = torch.empty((self.N, self.num_moves), dtype=torch.uint8, device=self.device)
        assert out.stride(-1) == 1 and out.shape[0] == self.N
        check(lib.hz_env_observe(self._h, self.mdp, out.data_ptr(), _OBS_DTYPES[out.dtype], out.stride(0),
                                 None if packed is None else packed.data_ptr(), legal.data_ptr(), _stream()),
              "hz_env_observe")
        return out, legal

    def observe_packed(self, packed, legal):
        check(lib.hz_env_observe(self._h, self.mdp, None, 0, 0, packed.data_ptr(), legal.data_ptr(), _stream()),
              "hz_env_observe")

    def snapshot(self):
        """(states [N, 32] i32, generator positions [N] i32) as of the work enqueued so far (include/hz_env.h::hz_env_snapshot)."""
        st = torch.empty((self.N, 32), dtype=torch.int32, device=self.device)
        pos = torch.empty(self.N, dtype=torch.int32, device=self.device)
        check(lib.hz_env_snapshot(self._h, st.data_ptr(), pos.data_ptr(), _stream()), "hz_env_snapshot")
        return st, pos

    def restore(self, snap):
        check(lib.hz_env_restore(self._h, snap[0].data_ptr(), snap[1].data_ptr(), _stream()), "hz_env_restore")

    def probe(self):
        out = torch.empty((self.N, 16), dtype=torch.int32, device=self.device)
        check(lib.hz_env_probe(self._h, out.data_ptr(), _stream()), "hz_env_probe")
        return out


class _Discrete:
    """gym.spaces.Discrete stand-in (rl_env.py:142 uses only .n)."""

    def __init__(self, n):
        self.n = n


class HanabiEnv:
    """envs/hanabi/rl_env.py HanabiEnv: args = {"hanabi_name": ..., "seed": ...}; reset() -> (share_obs, obs, legal);
    step(a) -> (share_obs, obs, reward, done, {'score'}, legal).  One game on the GPU."""

    def __init__(self, args):
        seed = 0 if args.get("seed") is None else args["seed"]  # rl_env.py:106-109
        self._g = HanabiVecEnv(args["hanabi_name"], [seed], mdp="global")
        self.players = self._g.players
        n = self._g.num_moves
        self.action_space = [_Discrete(n) for _ in range(self.players)]
        self.observation_space = [[self._g.obs_len + self.players] for _ in range(self.players)]
        self.share_observation_space = [[self._g.own_len + self._g.obs_len + self.players] for _ in range(self.players)]

    def num_moves(self):
        return self._g.num_moves

    def vectorized_observation_shape(self):
        return [self._g.obs_len]

    def vectorized_share_observation_shape(self):
        return [self._g.own_len + self._g.obs_len]

    def _obs(self):
        share, legal = self._g.observe()
        share = share[0].cpu().numpy().astype(np.int64)
        return share.tolist(), share[self._g.own_len:].tolist(), legal[0].cpu().numpy().astype(np.float64).tolist()

    def reset(self, choose=True):
        self._g.reset()
        return self._obs()

    def step(self, action):
        if not isinstance(action, (int, np.integer)):
            raise ValueError("Expected action as dict or int, got: {}".format(action))  # rl_env.py:414-416
        reward, done, score, status = self._g.step(torch.tensor([int(action)], dtype=torch.int32))
        vals = torch.stack((reward, done.int(), score, status)).cpu().numpy()[:, 0]
        if vals[3] != 0:
            raise AssertionError("illegal move %d" % int(action))  # reference aborts the process here
        share, obs, legal = self._obs()
        return share, obs, int(vals[0]), bool(vals[1]), {"score": int(vals[2])}, legal

    def close(self):
        pass


class HanabiControlWrapper:
    """config/hanabi_control/env_wrapper.py:6-34 (subclass of core.game.Game there)."""

    def __init__(self, env, discount, cvt_string=False, mdp="global"):
        self.env = env
        self.action_space_size = env.num_moves()
        self.discount = discount
        self.config = None
        self.cvt_string = cvt_string
        self.mdp = mdp
        env.action_space = env.action_space[0] if isinstance(env.action_space, list) else env.action_space

    def legal_actions(self):
        return [_ for _ in range(self.env.action_space.n)]

    def step(self, action):
        global_state, state, reward, done, info, legal = self.env.step(int(action))
        o = global_state if self.mdp == "global" else state
        return np.array(o), np.array(reward), np.array(done), np.array(info), np.array(legal)

    def reset(self, **kwargs):
        global_state, state, legal = self.env.reset()
        o = global_state if self.mdp == "global" else state
        return np.array(o), np.array(legal)

    def close(self):
        pass

    def render(self, *a, **k):
        pass

