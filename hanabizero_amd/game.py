"""hanabizero_amd.game -- ``Game`` / ``GameHistory`` behind the reference's API surface (/root/reference/core/game.py:26-214),
so that core/train.py, core/replay_buffer.py and core/reanalyze_worker.py consume self-play output unchanged.

The reference keeps a trajectory as seven Python lists that ``game_over()`` turns into arrays (and the observations into
a Ray ObjectRef).  Here a trajectory IS a set of arrays from the start -- one growable column per field (``_Column``) --
because its producer is the GPU actor, whose finished games arrive as whole arrays (``from_arrays`` / ``from_packed``:
the replay ingest format of SURVEY.md section 8f-2) and whose consumers slice it (``obs``, ``make_batch``).  The reference's
attribute names (``obs_history``, ``actions``, ``rewards``, ``child_visits``, ``root_values``, ``legal_actions``) are views
of the filled part of those columns, writable in place like the reference's arrays (``put()`` rewrites ``rewards``,
``store_search_stats(idx=...)`` rewrites a row), and the move-by-move methods of the reference (``init`` / ``append`` /
``store_search_stats`` / ``pad_over`` / ``step_obs`` / ``game_over``) append to them.  No Ray: ``obs()`` indexes the frames
directly.  Behaviour is pinned against the reference's own class by tests/golden/game_history.npz.
"""
import numpy as np


class Game:
    """What a config's ``new_game`` returns (core/game.py:26-46): an env wrapper with ``reset`` / ``step`` / ``close``."""

    def __init__(self, env, action_space_size, discount, config=None):
        self.env, self.action_space_size, self.discount, self.config = env, action_space_size, discount, config

    def legal_actions(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def close(self, *args, **kwargs):
        self.env.close(*args, **kwargs)

    def render(self, *args, **kwargs):
        self.env.render(*args, **kwargs)


class _Column:
    """A growable array of rows: amortised O(1) append, the filled part as a writable view."""

    __slots__ = ("buf", "n")

    def __init__(self, rows=None):
        self.buf, self.n = None, 0
        if rows is not None:
            self.buf = np.asarray(rows)
            self.n = len(self.buf)

    def push(self, row):
        row = np.asarray(row)
        if self.buf is None:
            self.buf = np.empty((16,) + row.shape, row.dtype)
        dtype = np.result_type(self.buf.dtype, row.dtype)  # (an int column that meets a float becomes a float column)
        full = self.n == len(self.buf)
        if full or dtype != self.buf.dtype:
            grown = np.empty((max(16, 2 * len(self.buf)) if full else len(self.buf),) + self.buf.shape[1:], dtype)
            grown[:self.n] = self.buf[:self.n]
            self.buf = grown
        self.buf[self.n] = row
        self.n += 1

    def extend(self, rows):
        for row in rows:
            self.push(row)

    @property
    def view(self):
        return np.empty((0,)) if self.buf is None else self.buf[:self.n]


class GameHistory:
    """One self-play trajectory (core/game.py:49-214).  T = len(self) moves: ``obs_history`` [stack + T (+ pad), D] (the
    first frame repeated ``stack`` times, then one frame per move), ``legal_actions`` [T + 1, A], ``actions`` / ``rewards``
    [T], ``child_visits`` [T, A] (visit counts normalised to sum 1), ``root_values`` [T]."""

    ks = ['visits', 'root', 'actions', 'obs', 'reward', 'tar_v', 'tar_r', 'tar_p']

    def __init__(self, action_space, max_length=200, config=None):
        self.action_space, self.max_length, self.config = action_space, max_length, config
        self.stacked_observations = config.stacked_observations
        self.discount = config.discount
        self.action_space_size = config.action_space_size
        self._reset_columns()

    def _reset_columns(self):
        self._cols = {k: _Column() for k in ("obs", "legal", "action", "reward", "visits", "value")}
        self.target_values, self.target_rewards, self.target_policies = [], [], []

    # -- the reference's attribute names: views of the columns' filled parts (assignment replaces a column) -----------
    def _get(name):
        return lambda self: self._cols[name].view

    def _set(name):
        def setter(self, rows):
            self._cols[name] = _Column(rows)
        return setter
    obs_history = property(_get("obs"), _set("obs"))
    legal_actions = property(_get("legal"), _set("legal"))
    actions = property(_get("action"), _set("action"))
    rewards = property(_get("reward"), _set("reward"))
    child_visits = property(_get("visits"), _set("visits"))
    root_values = property(_get("value"), _set("value"))
    del _get, _set

    # -- building a trajectory move by move (selfplay_worker.py:122-138, 300-330) -------------------------------------
    def init(self, init_observations, init_legal_action):
        assert len(init_observations) == self.stacked_observations
        self._reset_columns()
        self._cols["obs"].extend(init_observations)
        self._cols["legal"].push(init_legal_action)

    def store_search_stats(self, visit_counts, root_value, idx=None, set_flag=False):
        """Root visit counts, normalised by their sum, and the root value of the move about to be played; with `idx` the
        statistics of move idx are replaced (the reanalyze write-back of the reference; set_flag is its read-only toggle,
        which these arrays do not need)."""
        counts = np.asarray(visit_counts, np.float64)
        row = counts / counts.sum()
        if idx is None:
            self._cols["visits"].push(row)
            self._cols["value"].push(np.float64(root_value))
        else:
            self.child_visits[idx] = row
            self.root_values[idx] = root_value

    def append(self, action, obs, reward, legal_action):
        c = self._cols
        c["action"].push(action), c["obs"].push(obs), c["reward"].push(reward), c["legal"].push(legal_action)

    def pad_over(self, next_block_observations, next_block_rewards, next_block_root_values, next_block_child_visits,
                 next_legal_a):
        """The head of the next block of a split trajectory appended for the unroll / bootstrap windows that cross the cut."""
        cfg = self.config
        assert len(next_block_observations) <= cfg.num_unroll_steps and len(next_block_child_visits) <= cfg.num_unroll_steps
        assert len(next_block_root_values) <= cfg.num_unroll_steps + cfg.td_steps
        assert len(next_block_rewards) <= cfg.num_unroll_steps + cfg.td_steps - 1
        c = self._cols
        c["obs"].extend(next_block_observations), c["legal"].extend(next_legal_a), c["reward"].extend(next_block_rewards)
        c["value"].extend(next_block_root_values), c["visits"].extend(next_block_child_visits)

    def game_over(self):
        """The reference freezes its lists into arrays here; the columns already are arrays (trimmed to their length)."""
        for col in self._cols.values():
            if col.buf is not None:
                col.buf = col.buf[:col.n]

    def is_full(self):
        return len(self) >= self.max_length

    def __len__(self):
        return self._cols["action"].n

    # -- reading ----------------------------------------------------------------------------------------------------
    def obs(self, i, extra_len=0, padding=False):
        """Frames [i, i + stack + extra_len): the stacked window of position i plus `extra_len` successors; with `padding`
        a window that runs past the end repeats the last frame."""
        want = self.stacked_observations + extra_len
        frames = self.obs_history[i:i + want]
        if padding and len(frames) < want:
            frames = np.concatenate((frames, np.repeat(frames[-1:], want - len(frames), axis=0)))
        return frames

    def step_obs(self):
        """The window the next root inference sees: the `stack` newest frames."""
        n = self._cols["reward"].n
        return self.obs_history[n:n + self.stacked_observations]

    def zero_obs(self):
        return [np.zeros(self.config.obs_shape // self.stacked_observations) for _ in range(self.stacked_observations)]

    def obs_object(self):
        return self.obs_history

    def action_history(self, idx=None):
        return self.actions if idx is None else self.actions[:idx]

    def get_targets(self, i):
        return self.target_values[i], self.target_rewards[i], self.target_policies[i]

    # -- persistence (replay_buffer.save_files / load_files) --------------------------------------------------------
    def save_file(self):
        return {'vis': np.array(self.child_visits), 'root': np.array(self.root_values), 'a': np.array(self.actions),
                'o': np.array(self.obs_history), 'r': np.array(self.rewards), 'la': np.array(self.legal_actions),
                'tar_v': np.array(self.target_values), 'tar_r': np.array(self.target_rewards),
                'tar_p': np.array(self.target_policies)}

    def load_file(self, gdict):
        self.child_visits, self.root_values, self.actions = gdict['vis'], gdict['root'], gdict['a']
        self.obs_history, self.rewards, self.legal_actions = gdict['o'], gdict['r'], gdict['la']
        self.target_values, self.target_rewards, self.target_policies = gdict['tar_v'], gdict['tar_r'], gdict['tar_p']

    # -- replay ingest: whole arrays / packed GPU record -> history ---------------------------------------------------
    @classmethod
    def from_arrays(cls, action_space, config, actions, rewards, child_visits, root_values, legal_actions, frames):
        """A finished history from whole arrays: what init() + T x (store_search_stats, append) + game_over() leave behind,
        without the move-by-move Python.  frames [T+1, D] (0/1; kept in the dtype given -- uint8 from the packed records,
        where the reference holds int64 lists), legal_actions [T+1, A] float64, child_visits [T, A] float64 (already
        normalised), root_values [T] float64, actions / rewards [T] int64."""
        g = cls(action_space, max_length=config.history_length, config=config)
        stack = config.stacked_observations
        frames = np.asarray(frames)
        g.obs_history = np.concatenate((np.repeat(frames[:1], stack - 1, axis=0), frames), axis=0) if stack > 1 else frames
        g.actions, g.rewards, g.child_visits, g.root_values, g.legal_actions = actions, rewards, child_visits, root_values, legal_actions
        return g

    @staticmethod
    def _unpack_frames(rec, config):
        T = int(rec["len"])
        D = config.obs_shape // config.stacked_observations
        return np.unpackbits(np.ascontiguousarray(rec["obs_bits"][:T + 1]).view(np.uint8), axis=1, bitorder="little")[:, :D]

    @classmethod
    def from_packed(cls, rec, action_space, config):
        """rec: dict produced by hanabizero_amd.selfplay.unpack_record (numpy arrays of ONE finished game):
        len, action [T], reward [T], visits [T, A] (masked counts), value [T], legal [T+1, A], obs_bits [T+1, W] u32.
        Equals the history the actor would build move by move (`from_packed_stepwise`; tests/test_game.py compares the
        two); rewards are the raw env rewards (reshape_turn_rewards = the reference's put() applies to the result)."""
        T = int(rec["len"])
        counts = rec["visits"][:T].astype(np.int64)
        return cls.from_arrays(action_space, config, rec["action"][:T].astype(np.int64), rec["reward"][:T].astype(np.int64),
                               counts / counts.sum(1, keepdims=True), rec["value"][:T].astype(np.float64),
                               rec["legal"][:T + 1].astype(np.float64), cls._unpack_frames(rec, config))

    @classmethod
    def from_packed_stepwise(cls, rec, action_space, config):
        """The same history through the move-by-move interface (the specification of from_packed; used by the tests)."""
        g = cls(action_space, max_length=config.history_length, config=config)
        T = int(rec["len"])
        frames = cls._unpack_frames(rec, config).astype(np.int64)
        legal = rec["legal"][:T + 1].astype(np.float64)
        g.init([frames[0]] * config.stacked_observations, legal[0])
        for t in range(T):
            g.store_search_stats([int(c) for c in rec["visits"][t]], float(rec["value"][t]))
            g.append(int(rec["action"][t]), frames[t + 1], int(rec["reward"][t]), legal[t + 1])
        g.game_over()
        return g


def reshape_turn_rewards(game_history):
    """DataWorker.put (selfplay_worker.py:29-39): a move's reward becomes the sum of its own and the previous move's raw
    rewards -- the return of one full turn of the two players -- r'[t] = r[t] + r[t-1] (r'[0] = r[0]), in place."""
    r = np.asarray(game_history.rewards)
    shifted = r[:-1].copy()
    if isinstance(game_history.rewards, np.ndarray):
        game_history.rewards[1:] += shifted
    else:  # a plain list (histories that are not GameHistory objects)
        for t in range(1, len(r)):
            game_history.rewards[t] = r[t] + shifted[t - 1]
    return game_history
