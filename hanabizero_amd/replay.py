"""hanabizero_amd.replay -- the prioritised replay buffer of HanabiZero without Ray.

What it replaces: ``ReplayBuffer`` (/root/reference/core/replay_buffer.py:11-215), the Ray actor the reference's
DataWorkers push finished games into (``save_pools.remote``, selfplay_worker.py:75-78) and its batch makers sample from
(``prepare_batch_context.remote``, reanalyze_worker.py:205-206).  Same bookkeeping: one priority per stored position
(new games enter at the current maximum priority when ``use_max_priority``: replay_buffer.py:116-118), sampling without
replacement with probabilities ``priority ** alpha`` and importance weights ``(N p) ** -beta / max`` (:140-166), priority
write-back (:174-178), oldest games dropped beyond ``transition_top`` positions (:180-215).
``ingest_packed`` is the new front door: the byte buffers hanabizero_amd.dist.gather_packed lands on the replay owner
become reference-shaped ``GameHistory`` objects (turn-reward reshape of DataWorker.put included)."""
import numpy as np

from .game import GameHistory, reshape_turn_rewards
from .selfplay import unpack_packed, unpack_record


class ReplayBuffer:
    def __init__(self, config, priority_prob_alpha=0.6, uniform_ratio=0.0, transition_top=None, seed=0):
        self.config = config
        self.batch_size = config.batch_size
        self.buffer, self.game_look_up = [], []
        self.priorities = np.zeros(0, np.float64)
        self.base_idx = 0
        self._alpha = priority_prob_alpha if getattr(config, "use_priority", True) else 0.0  # core/config.py:166, 267
        self.uniform_ratio = uniform_ratio
        self.transition_top = int(transition_top if transition_top is not None else 25 * 100 * 10 ** 4)  # :38
        self.keep_ratio = 1
        self._eps_collected = 0
        self.clear_time = 0
        self.rng = np.random.RandomState(seed)

    # -- ingest ------------------------------------------------------------------------------------------------
    def save_pools(self, pools, gap_step=0):
        for game, priorities in pools:
            self.save_game(game, True, gap_step, priorities)

    def save_game(self, game, end_tag, gap_steps, priorities=None):  # replay_buffer.py:108-132
        valid_len = len(game) if end_tag else len(game) - gap_steps
        if end_tag:
            self._eps_collected += 1
        if priorities is None:
            max_prio = self.priorities.max() if self.buffer else 1
            new = [max_prio] * valid_len + [0.0] * (len(game) - valid_len)
        else:
            assert len(game) == len(priorities), "priorities should be of same length as the game steps"
            new = np.asarray(priorities, np.float64).reshape(-1)
        self.priorities = np.concatenate((self.priorities, new))
        self.buffer.append(game)
        self.game_look_up += [(self.base_idx + len(self.buffer) - 1, pos) for pos in range(len(game))]

    def ingest_packed(self, buf, n, moves, action_space=None):
        """All games of one packed byte buffer (SelfPlayActor.drain_packed / dist.gather_packed) -> GameHistory objects
        (the histories GameHistory.from_packed builds), turn rewards reshaped as DataWorker.put does
        (selfplay_worker.py:32-37), stored at the maximum priority (--use_max_priority, train.sh).  Whole-buffer numpy:
        one bit-unpack, one normalisation of the visit counts, one priority append and one look-up extension per buffer;
        the per-game work is slicing.  Returns the number of games."""
        cfg = self.config
        A, D = cfg.action_space_size, cfg.obs_shape // cfg.stacked_observations
        rec = unpack_packed(buf, n, moves, A, (D + 31) // 32)
        lens = rec["meta"][:, 0].astype(np.int64)
        start = rec["start"]
        frames = np.unpackbits(np.ascontiguousarray(rec["obs"]).view(np.uint8), axis=1, bitorder="little")[:, :D]
        counts = rec["visits"].astype(np.int64)
        visits = counts / counts.sum(1, keepdims=True)
        values = rec["value"].astype(np.float64)
        legal = rec["legal"].astype(np.float64)
        actions = rec["action"].astype(np.int64)
        raw = rec["reward"].astype(np.int64)
        rewards = raw.copy()                      # DataWorker.put: r'[t] = r[t] + r[t-1] inside a game
        rewards[1:] += raw[:-1]
        rewards[start] = raw[start]
        max_prio = self.priorities.max() if self.buffer else 1
        first = self.base_idx + len(self.buffer)
        for i in range(n):
            s, T = int(start[i]), int(lens[i])
            self.buffer.append(GameHistory.from_arrays(action_space, cfg, actions[s:s + T], rewards[s:s + T], visits[s:s + T],
                                                       values[s:s + T], legal[s + i:s + i + T + 1], frames[s + i:s + i + T + 1]))
        self._eps_collected += n
        self.priorities = np.concatenate((self.priorities, np.full(int(lens.sum()), float(max_prio))))
        self.game_look_up += [(first + i, pos) for i in range(n) for pos in range(int(lens[i]))]
        return n

    # -- sampling ------------------------------------------------------------------------------------------------
    def get_total_len(self):
        return len(self.priorities)

    def size(self):
        return len(self.buffer)

    def episodes_collected(self):
        return self._eps_collected

    def prepare_batch_context(self, batch_size, beta):  # replay_buffer.py:140-172
        assert beta > 0
        total = self.get_total_len()
        assert total > batch_size, "not enough positions (%d) for a batch of %d" % (total, batch_size)
        alpha = 0.0 if self.rng.random_sample() < self.uniform_ratio else self._alpha
        probs = self.priorities ** alpha
        probs = probs / probs.sum()
        indices = self.rng.choice(total, batch_size, p=probs, replace=False)
        weights = (total * probs[indices]) ** (-beta)
        weights = weights / weights.max()
        games, positions = [], []
        for idx in indices:
            game_id, pos = self.game_look_up[idx]
            games.append(self.buffer[game_id - self.base_idx])
            positions.append(pos)
        # stamped with the eviction epoch it was drawn in (the reference stamps wall-clock time on both sides,
        # replay_buffer.py:171, 205): remove_to_fit shifts every index, so write-backs of older batches must be dropped
        return games, positions, indices, weights.astype(np.float32), [self.clear_time + 1] * batch_size

    def update_priorities(self, batch_indices, batch_priorities, make_time=None):
        """Priority write-back of a batch drawn by prepare_batch_context; entries drawn before the latest eviction
        (make_time <= clear_time) are ignored, as in the reference (replay_buffer.py:174-178)."""
        for i in range(len(batch_indices)):
            if make_time is None or make_time[i] > self.clear_time:
                self.priorities[batch_indices[i]] = batch_priorities[i]

    def remove_to_fit(self):  # replay_buffer.py:180-215
        total = self.get_total_len()
        if total <= self.transition_top:
            return 0
        index = 0
        for i in range(self.size()):
            total -= len(self.buffer[i])
            if total <= self.transition_top * self.keep_ratio:
                index = i
                break
        if total < self.batch_size:
            return 0
        excess = index + 1
        steps = sum(len(g) for g in self.buffer[:excess])
        del self.buffer[:excess]
        self.priorities = self.priorities[steps:]
        del self.game_look_up[:steps]
        self.base_idx += excess
        self.clear_time = self.clear_time + 1
        return excess
