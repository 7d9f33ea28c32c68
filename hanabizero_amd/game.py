"""hanabizero_amd.game -- ``Game`` / ``GameHistory`` with the reference's API surface (/root/reference/core/game.py:26-214)
so that core/train.py, core/replay_buffer.py and core/reanalyze_worker.py can consume self-play output unchanged.

Differences: no Ray (``obs_history`` is a plain ndarray after ``game_over()`` instead of a plasma ObjectRef,
game.py:181; ``obs()`` therefore indexes it directly, game.py:153), and ``GameHistory.from_packed`` rebuilds a
history from the fixed-layout record the GPU actor emits (hanabizero_amd/selfplay.py) -- the replay ingest format of
SURVEY.md section 8f-2.
"""
import copy

import numpy as np


class Game:  # core/game.py:26-46
    def __init__(self, env, action_space_size, discount, config=None):
        self.env = env
        self.action_space_size = action_space_size
        self.discount = discount
        self.config = config

    def legal_actions(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError()

    def close(self, *args, **kwargs):
        self.env.close(*args, **kwargs)

    def render(self, *args, **kwargs):
        self.env.render(*args, **kwargs)


class GameHistory:  # core/game.py:49-214
    def __init__(self, action_space, max_length=200, config=None):
        self.action_space = action_space
        self.max_length = max_length
        self.config = config
        self.stacked_observations = config.stacked_observations
        self.discount = config.discount
        self.action_space_size = config.action_space_size
        self.child_visits, self.root_values = [], []
        self.actions, self.obs_history, self.rewards = [], [], []
        self.legal_actions = []
        self.ks = ['visits', 'root', 'actions', 'obs', 'reward', 'tar_v', 'tar_r', 'tar_p']

    def init(self, init_observations, init_legal_action):
        self.child_visits, self.root_values = [], []
        self.actions, self.obs_history, self.rewards = [], [], []
        self.target_values, self.target_rewards, self.target_policies = [], [], []
        self.legal_actions = []
        assert len(init_observations) == self.stacked_observations
        for observation in init_observations:
            self.obs_history.append(copy.deepcopy(observation))
        self.legal_actions.append(init_legal_action)

    def pad_over(self, next_block_observations, next_block_rewards, next_block_root_values, next_block_child_visits,
                 next_legal_a):
        assert len(next_block_observations) <= self.config.num_unroll_steps
        assert len(next_block_child_visits) <= self.config.num_unroll_steps
        assert len(next_block_root_values) <= self.config.num_unroll_steps + self.config.td_steps
        assert len(next_block_rewards) <= self.config.num_unroll_steps + self.config.td_steps - 1
        for observation in next_block_observations:
            self.obs_history.append(copy.deepcopy(observation))
        for la in next_legal_a:
            self.legal_actions.append(copy.deepcopy(la))
        for reward in next_block_rewards:
            self.rewards.append(reward)
        for value in next_block_root_values:
            self.root_values.append(value)
        for child_visits in next_block_child_visits:
            self.child_visits.append(child_visits)

    def is_full(self):
        return self.__len__() >= self.max_length

    def load_file(self, gdict):
        self.target_values, self.target_rewards, self.target_policies = gdict['tar_v'], gdict['tar_r'], gdict['tar_p']
        self.child_visits, self.root_values = gdict['vis'], gdict['root']
        self.actions, self.obs_history, self.rewards = gdict['a'], gdict['o'], gdict['r']
        self.legal_actions = gdict['la']

    def save_file(self):
        return {'vis': np.array(self.child_visits), 'root': np.array(self.root_values), 'a': np.array(self.actions),
                'o': np.array(self.obs_history), 'r': np.array(self.rewards), 'tar_v': np.array(self.target_values),
                'tar_r': np.array(self.target_rewards), 'tar_p': np.array(self.target_policies),
                'la': np.array(self.legal_actions)}

    def append(self, action, obs, reward, legal_action):
        self.actions.append(action)
        self.obs_history.append(obs)
        self.rewards.append(reward)
        self.legal_actions.append(legal_action)

    def obs_object(self):
        return self.obs_history

    def obs(self, i, extra_len=0, padding=False):
        frames = self.obs_history[i:i + self.stacked_observations + extra_len]
        if padding:
            pad_len = self.stacked_observations + extra_len - len(frames)
            if pad_len > 0:
                pad_frames = [frames[-1] for _ in range(pad_len)]
                frames = np.concatenate((frames, pad_frames))
        return frames

    def zero_obs(self):
        return [np.zeros(self.config.obs_shape // self.stacked_observations) for _ in range(self.stacked_observations)]

    def step_obs(self):
        index = len(self.rewards)
        return self.obs_history[index:index + self.stacked_observations]

    def get_targets(self, i):
        return self.target_values[i], self.target_rewards[i], self.target_policies[i]

    def game_over(self):
        self.rewards = np.array(self.rewards)
        self.obs_history = np.array(self.obs_history)
        self.actions = np.array(self.actions)
        self.child_visits = np.array(self.child_visits)
        self.root_values = np.array(self.root_values)
        self.legal_actions = np.array(self.legal_actions)

    def store_search_stats(self, visit_counts, root_value, idx=None, set_flag=False):
        if set_flag:
            self.child_visits.setflags(write=1)
            self.root_values.setflags(write=1)
        sum_visits = sum(visit_counts)
        if idx is None:
            self.child_visits.append([visit_count / sum_visits for visit_count in visit_counts])
            self.root_values.append(root_value)
        else:
            self.child_visits[idx] = [visit_count / sum_visits for visit_count in visit_counts]
            self.root_values[idx] = root_value
        if set_flag:
            self.child_visits.setflags(write=0)
            self.root_values.setflags(write=0)

    def action_history(self, idx=None):
        return self.actions if idx is None else self.actions[:idx]

    def __len__(self):
        return len(self.actions)

    # -- replay ingest: packed GPU record -> reference-shaped history --------------------------------------
    @classmethod
    def from_arrays(cls, action_space, config, actions, rewards, child_visits, root_values, legal_actions, frames):
        """A finished history from whole arrays: what init() + T x (store_search_stats, append) + game_over() leave behind
        (selfplay_worker.py:216-228, 300-308), without the move-by-move Python.  frames [T+1, D] (0/1; kept in the dtype
        given -- uint8 from the packed records, where the reference holds int64 lists), legal_actions [T+1, A] float64,
        child_visits [T, A] float64 (already normalised), root_values [T] float64, actions / rewards [T] int64."""
        g = cls(action_space, max_length=config.history_length, config=config)
        g.target_values, g.target_rewards, g.target_policies = [], [], []
        stack = config.stacked_observations
        g.obs_history = np.concatenate((np.repeat(frames[:1], stack - 1, axis=0), frames), axis=0) if stack > 1 else np.asarray(frames)
        g.actions, g.rewards = actions, rewards
        g.child_visits, g.root_values = child_visits, root_values
        g.legal_actions = legal_actions
        return g

    @classmethod
    def from_packed(cls, rec, action_space, config):
        """rec: dict produced by hanabizero_amd.selfplay.unpack_record (numpy arrays of ONE finished game):
        len, action [T], reward [T], visits [T, A] (masked counts), value [T], legal [T+1, A], obs_bits [T+1, W] u32.
        Equivalent to the history the reference actor builds move by move and then closes with game_over()
        (`from_packed_stepwise` below does exactly that; tests/test_game.py compares the two); rewards are the raw env
        rewards (put() reshapes them later)."""
        T = int(rec["len"])
        D = config.obs_shape // config.stacked_observations
        frames = np.unpackbits(np.ascontiguousarray(rec["obs_bits"][:T + 1]).view(np.uint8), axis=1, bitorder="little")[:, :D]
        counts = rec["visits"][:T].astype(np.int64)
        return cls.from_arrays(action_space, config, rec["action"][:T].astype(np.int64), rec["reward"][:T].astype(np.int64),
                               counts / counts.sum(1, keepdims=True), rec["value"][:T].astype(np.float64),
                               rec["legal"][:T + 1].astype(np.float64), frames)

    @classmethod
    def from_packed_stepwise(cls, rec, action_space, config):
        """The same history built the way the reference actor builds it, one move at a time (the specification of
        from_packed; used by the tests)."""
        g = cls(action_space, max_length=config.history_length, config=config)
        T = int(rec["len"])
        D = config.obs_shape // config.stacked_observations
        bits = np.unpackbits(np.ascontiguousarray(rec["obs_bits"][:T + 1]).view(np.uint8), axis=1, bitorder="little")[:, :D]
        frames = bits.astype(np.int64)
        legal = rec["legal"][:T + 1].astype(np.float64)
        g.init([frames[0] for _ in range(config.stacked_observations)], legal[0])
        for t in range(T):
            counts = [int(c) for c in rec["visits"][t]]
            g.store_search_stats(counts, float(rec["value"][t]))
            g.append(int(rec["action"][t]), frames[t + 1], int(rec["reward"][t]), legal[t + 1])
        g.game_over()
        return g


def reshape_turn_rewards(game_history):
    """DataWorker.put (selfplay_worker.py:32-37): r'[t] = r[t] + r[t-1] with the ORIGINAL r[t-1], in place."""
    r = game_history.rewards
    if isinstance(r, np.ndarray):
        if len(r) > 1:
            r[1:] = r[1:] + r[:-1].copy()
        return game_history
    prev_r = r[0]
    for step_id in range(1, len(r)):
        cur_r = r[step_id] + prev_r
        prev_r = r[step_id]
        r[step_id] = cur_r
    return game_history
