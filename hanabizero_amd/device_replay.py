"""hanabizero_amd.device_replay -- the prioritised replay and the learner's batch maker resident in HBM.

What it replaces: the host side of the reference's batch pipeline -- ``ReplayBuffer`` (/root/reference/core/replay_buffer.py:
92-215), ``BatchWorker_CPU`` (core/reanalyze_worker.py:45-168: sampling contexts, frame stacking, random padding actions) and
the non-searching half of ``BatchWorker_GPU`` (:249-304 value / reward targets, :374-399 stored policy targets) -- which in
the reference are Ray actors exchanging float32 frame stacks through the object store.  `hanabizero_amd.replay.ReplayBuffer` +
`hanabizero_amd.learner.make_batch` restate that on the host (pinned by fixtures recorded from the reference); this module is
the MI355X form of the same arithmetic:

  * the replay IS the actors' packed records (include/hz_selfplay.h hz_actor_pack), appended section by section to flat arrays
    in HBM: per position action / turn-reshaped reward / root value / visit counts / (t, T, first frame row of its game), per
    frame the legal mask and the observation as 32-bit words (44 words = 176 B for a Hanabi-Full 5p frame instead of 5.5 KB of
    float32): ~330 B per position, so the reference's 25 M-position window is 8 GB of the 288;
  * a batch is index arithmetic on those arrays: prioritised sampling without replacement (exponential keys + top-k, the
    distribution of numpy's sequential draw), window gather + bit expansion by `hz_replay_windows` (include/hz_replay.h) in
    the consumer's element type, td-step returns in float64 with the terms added in the reference's order, visit counts
    normalised -- no host work per sample, no PCIe traffic, no synchronisation: the learner's static input tensors are written
    in place and the step's hipGraph replays behind them.

tests/test_device_replay.py holds `assemble()` to `learner.make_batch` (bit-equal targets and inputs on the same sampled
positions) and `ingest` to `ReplayBuffer.ingest_packed`.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib
from .selfplay import packed_layout

_TORCH_OF = {np.dtype(np.int8): torch.int8, np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32,
             np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8}
_OBS_CODE = {torch.float32: 1, torch.bfloat16: 2, torch.float16: 3}  # include/hz_env.h HZ_OBS_*


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceReplay:
    def __init__(self, config, capacity, device=None, priority_prob_alpha=0.6, transition_top=None, games_capacity=None, seed=0):
        """capacity: positions the arrays hold (>= transition_top + what arrives between two remove_to_fit calls);
        games_capacity: frame rows beyond one per position (a game of T moves has T + 1 frames), default capacity // 4."""
        self.config = config
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        d = self.device
        self.A = config.action_space_size
        self.stack = config.stacked_observations
        self.D = config.obs_shape // self.stack
        self.W = (self.D + 31) // 32
        self.P = int(capacity)
        self.F = self.P + int(games_capacity if games_capacity is not None else max(1024, self.P // 4))
        self.alpha = priority_prob_alpha if getattr(config, "use_priority", True) else 0.0
        self.transition_top = int(transition_top if transition_top is not None else 25 * 100 * 10 ** 4)  # replay_buffer.py:38
        z = lambda *s, dtype: torch.zeros(s, dtype=dtype, device=d)
        P, F, A, W = self.P, self.F, self.A, self.W
        self.action, self.reward = z(P, dtype=torch.int8), z(P, dtype=torch.int16)
        self.value, self.visits = z(P, dtype=torch.float32), z(P, A, dtype=torch.int16)
        self.pos_t, self.pos_T, self.pos_row0 = z(P, dtype=torch.int32), z(P, dtype=torch.int32), z(P, dtype=torch.int64)
        self.priority = z(P + 1, dtype=torch.float64)  # (entry P: where write-backs for evicted positions land)
        self.legal, self.frames = z(F, A, dtype=torch.uint8), z(F, W, dtype=torch.int32)
        self.tail = self.head = 0        # live positions [tail, head) (physical)
        self.ftail = self.fhead = 0      # live frame rows
        self.origin = 0                  # logical id of physical position 0 (ids handed out survive compaction)
        self.games = 0                   # games ingested so far
        self.gen = torch.Generator(device=d)
        self.gen.manual_seed(int(seed))
        self._gpow = {}                  # (discount, td_steps) -> discount ** i, i = 0 .. td_steps, float64 on the device

    # -- bookkeeping ----------------------------------------------------------------------------------------------
    def get_total_len(self):
        return self.head - self.tail

    def episodes_collected(self):
        return self.games

    @property
    def hbm_bytes(self):
        return sum(t.numel() * t.element_size() for t in (self.action, self.reward, self.value, self.visits, self.pos_t, self.pos_T,
                                                          self.pos_row0, self.priority, self.legal, self.frames))

    def _compact(self):
        """Live data to the front of the arrays (called when an ingest would run past their end)."""
        n, fn = self.head - self.tail, self.fhead - self.ftail
        if self.tail:
            for a in (self.action, self.reward, self.value, self.visits, self.pos_t, self.pos_T, self.pos_row0, self.priority):
                a[:n] = a[self.tail:self.head].clone()
            for a in (self.legal, self.frames):
                a[:fn] = a[self.ftail:self.fhead].clone()
            self.pos_row0[:n] -= self.ftail
            self.origin += self.tail
            self.tail, self.head, self.ftail, self.fhead = 0, n, 0, fn

    # -- ingest: one packed buffer of finished games, device to device ------------------------------------------------------
    def ingest_packed(self, buf, n, moves):
        """buf: uint8 tensor (device, or host: copied once) in the format of SelfPlayActor.drain_packed / dist.gather_packed --
        n games, `moves` moves in total.  Equals ReplayBuffer.ingest_packed (turn-reward reshape of DataWorker.put,
        selfplay_worker.py:32-37; new positions enter at the current maximum priority, replay_buffer.py:116-118).  No host
        loop over games; a device buffer is read on the CALLING stream without synchronisation (a receive buffer that the next gather
        reuses must not be rewritten before this stream has passed: tools/loop_bench.py orders the two with an event)."""
        n, moves = int(n), int(moves)
        if n == 0:
            return 0
        if not isinstance(buf, torch.Tensor):
            buf = torch.from_numpy(np.ascontiguousarray(buf))
        if buf.is_cuda:
            buf.record_stream(torch.cuda.current_stream(self.device))  # (the caller may drop its reference before this stream has read it)
        else:
            buf = buf.to(self.device)  # (a host buffer -- e.g. dist.gather_packed's reused landing memory -- is copied before this returns)
        if self.head + moves > self.P or self.fhead + moves + n > self.F:
            self.remove_to_fit(room=moves)
            self._compact()
            if self.head + moves > self.P or self.fhead + moves + n > self.F:
                raise RuntimeError("DeviceReplay: %d positions / %d frame rows do not fit (capacity %d / %d, %d / %d live)"
                                   % (moves, moves + n, self.P, self.F, self.head - self.tail, self.fhead - self.ftail))
        layout, total = packed_layout(n, moves, self.A, self.W)
        assert buf.numel() >= total
        sec = {k: buf[off:off + int(np.prod(shp)) * np.dtype(dt).itemsize].view(_TORCH_OF[np.dtype(dt)]).reshape(shp)
               for k, shp, dt, off in layout}
        d = self.device
        lens = sec["meta"][:, 0].to(torch.int64)
        start = torch.cumsum(lens, 0) - lens
        game = torch.repeat_interleave(torch.arange(n, device=d), lens, output_size=moves)
        h, fh = self.head, self.fhead
        s = slice(h, h + moves)
        self.pos_t[s] = (torch.arange(moves, device=d) - start[game]).to(torch.int32)
        self.pos_T[s] = lens[game].to(torch.int32)
        self.pos_row0[s] = (fh + start + torch.arange(n, device=d))[game]
        raw = sec["reward"].to(torch.int16)
        r = raw.clone()                      # DataWorker.put: r'[t] = r[t] + r[t - 1] inside a game
        r[1:] += raw[:-1]
        r[start] = raw[start]
        self.reward[s] = r
        self.action[s], self.value[s], self.visits[s] = sec["action"], sec["value"], sec["visits"]
        fs = slice(fh, fh + moves + n)
        self.legal[fs], self.frames[fs] = sec["legal"], sec["obs"]
        live = self.priority[self.tail:h]
        self.priority[s] = live.max() if h > self.tail else 1.0
        self.head, self.fhead, self.games = h + moves, fh + moves + n, self.games + n
        return n

    def remove_to_fit(self, room=0):
        """Drop the oldest whole games until at most transition_top (minus `room`) positions remain (replay_buffer.py:180-215;
        the reference calls it every 200 learner steps, train.py:367-368).  One small read-back; returns positions dropped."""
        total = self.head - self.tail
        top = min(self.transition_top, self.P - room)
        if total <= top:
            return 0
        first = self.head - top            # oldest position that may stay; the cut moves up to the next game start
        t = self.pos_t[first:self.head]
        starts = torch.nonzero(t == 0)
        if starts.numel() == 0:
            return 0
        cut = first + int(starts[0])
        dropped = cut - self.tail
        self.ftail = int(self.pos_row0[cut])
        self.tail = cut
        return dropped

    # -- sampling (replay_buffer.py:140-172) --------------------------------------------------------------------------------
    def sample(self, batch_size, beta):
        """(ids [B] int64 logical position ids, weights [B] float32) on the device: P(i) ~ priority_i ** alpha without
        replacement (exponential keys, the B largest: the distribution of a sequential weighted draw), importance weights
        (N P(i)) ** -beta / max."""
        total = self.head - self.tail
        assert total > batch_size, "not enough positions (%d) for a batch of %d" % (total, batch_size)
        probs = self.priority[self.tail:self.head] ** self.alpha
        probs = probs / probs.sum()
        keys = probs / torch.empty_like(probs).exponential_(1.0, generator=self.gen)
        idx = _topk_indices(keys, batch_size)
        w = (total * probs[idx]) ** (-beta)
        return idx + (self.tail + self.origin), (w / w.max()).to(torch.float32)

    def update_priorities(self, ids, priorities):
        """Write-back for a batch drawn by sample(); ids that have been evicted meanwhile are dropped (replay_buffer.py:174-178)."""
        phys = ids - self.origin
        self.priority[torch.where(phys >= self.tail, phys, torch.full_like(phys, self.P))] = priorities.to(torch.float64)

    # -- windows ------------------------------------------------------------------------------------------------------------
    def windows(self, phys, shift, valid, out, slot_elems=None):
        """out[m] = the stacked-observation window that ENDS at frame pos_t[phys[m]] + shift[m] of that position's game (zeros where
        not valid[m]), expanded from the packed frames into `out` ([M, >= stack * slot_elems], float32 / bf16 / fp16)."""
        M = phys.numel()
        t = torch.where(valid, self.pos_t[phys] + shift, torch.full_like(shift, -1)).to(torch.int32).contiguous()
        row0 = self.pos_row0[phys].contiguous()
        slot = self.D if slot_elems is None else int(slot_elems)
        assert out.shape[0] == M and out.stride(1) == 1
        check(lib.hz_replay_windows(self.frames.data_ptr(), self.W, row0.data_ptr(), t.data_ptr(), M, self.stack, self.D,
                                    out.data_ptr(), out.stride(0), slot, _OBS_CODE[out.dtype], _stream()), "hz_replay_windows")
        return out

    def windows_seq(self, phys, G, shift0, out, slot_elems=None, legal_out=None, valid_out=None):
        """windows() for B x G rows addressed through the position arrays (include/hz_replay.h hz_replay_windows_seq): row b * G + j =
        the window `shift0 + j` moves behind position phys[b], a zero row where the game has no such position; optionally the legal
        moves there and the validity mask.  One launch."""
        B = phys.numel()
        slot = self.D if slot_elems is None else int(slot_elems)
        assert out.shape[0] == B * G and out.stride(1) == 1 and phys.dtype == torch.int64 and phys.is_contiguous()
        check(lib.hz_replay_windows_seq(self.frames.data_ptr(), self.W, self.pos_row0.data_ptr(), self.pos_t.data_ptr(), self.pos_T.data_ptr(),
                                        phys.data_ptr(), B, G, int(shift0), self.stack, self.D, out.data_ptr(), out.stride(0), slot,
                                        _OBS_CODE[out.dtype], self.legal.data_ptr(), self.A, max(1, self.fhead),
                                        None if legal_out is None else legal_out.data_ptr(), None if valid_out is None else valid_out.data_ptr(),
                                        _stream()), "hz_replay_windows_seq")
        return out

    # -- one learner batch (reanalyze_worker.py:148-168, 249-304, 374-399) ------------------------------------------------------
    def assemble(self, ids, value_fn, out, rand_actions=None, value_windows=None, slot_elems=None):
        """Fills `out` -- an object with the learner's static input tensors: obs [B, stack, D] f32, action [B, U] int64,
        target_reward [B, U], target_value [B, U + 1], target_policy [B, U + 1, A] (hanabizero_amd.learner.GraphedUpdate has
        them) -- for the sampled position ids, exactly as learner.make_batch does on the host:
          obs            the window at the position (first frame repeated in front of a game's start)
          action         the U actions from the position on, uniformly random ones past the end of the game
                         (`rand_actions` [B, U] int64 if given: tests; drawn on the device otherwise)
          target_value   td_steps-step return: sum_i discount**i * reward[pos + k + i] (+0 past the end) + discount**td *
                         value_fn(window td_steps ahead) where that window exists; 0 past the end          (float64, cast once)
          target_reward  reward[pos + k], 0 past the end
          target_policy  visit counts / their sum, 0 past the end
        value_fn(windows [B * (U + 1), stack * slot] of value_windows.dtype) -> values [B * (U + 1)] (device tensor): the target
        model; `value_windows`: a caller-owned buffer for them (engine dtype, slots padded to `slot_elems`).  Returns the mask
        [B, U + 1] of unroll positions inside their games."""
        cfg = self.config
        U, td, A, g = cfg.num_unroll_steps, cfg.td_steps, self.A, cfg.discount
        d = self.device
        phys = (ids - self.origin).contiguous()
        B = phys.numel()
        # model input; bootstrap windows td steps ahead of every unroll position, the target model's values of them
        self.windows_seq(phys, 1, 0, out.obs.view(B, -1))
        if rand_actions is None:
            rand_actions = torch.randint(0, A, (B, U), device=d, generator=self.gen)
        slot = self.D if slot_elems is None else int(slot_elems)
        if value_windows is None:
            value_windows = torch.empty(B * (U + 1), self.stack * slot, dtype=torch.float32, device=d)
        self.windows_seq(phys, U + 1, td, value_windows, slot_elems=slot)
        boot = value_fn(value_windows).to(torch.float32).contiguous()
        # actions, reward / value / policy targets: one launch over the replay's arrays (include/hz_replay.h hz_replay_targets)
        gp = self._gpow.get((g, td))
        if gp is None:
            gp = self._gpow[(g, td)] = torch.tensor([g ** i for i in range(td + 1)], dtype=torch.float64, device=d)
        inside = torch.empty((B, U + 1), dtype=torch.bool, device=d)
        for t, dt in ((out.action, torch.int64), (out.target_reward, torch.float32), (out.target_value, torch.float32), (out.target_policy, torch.float32)):
            assert t.dtype == dt and t.is_contiguous()
        assert out.action.shape == (B, U) and out.target_reward.shape == (B, U) and out.target_value.shape == (B, U + 1) and \
            out.target_policy.shape == (B, U + 1, A) and rand_actions.shape == (B, U) and rand_actions.dtype == torch.int64 and boot.numel() == B * (U + 1)
        check(lib.hz_replay_targets(phys.data_ptr(), B, U, td, A, self.head, self.pos_t.data_ptr(), self.pos_T.data_ptr(), self.action.data_ptr(),
                                    self.reward.data_ptr(), self.visits.data_ptr(), boot.data_ptr(), gp.data_ptr(), rand_actions.contiguous().data_ptr(),
                                    out.action.data_ptr(), out.target_reward.data_ptr(), out.target_value.data_ptr(), out.target_policy.data_ptr(),
                                    inside.data_ptr(), _stream()), "hz_replay_targets")
        return inside

    # -- reanalyze context (reanalyze_worker.py:101-144) on the device ---------------------------------------------------------
    def policy_re_inputs(self, ids, out_windows, slot_elems=None):
        """For the positions `ids` [R]: the windows of their U + 1 unroll positions into `out_windows` [R * (U + 1), ...] (zero rows
        past the end of the game) plus (legal [R * (U + 1), A] uint8, all-zero past the end; mask [R * (U + 1)] bool)."""
        U = self.config.num_unroll_steps
        phys = (ids - self.origin).contiguous()
        R = phys.numel()
        legal = torch.empty((R * (U + 1), self.A), dtype=torch.uint8, device=self.device)
        mask = torch.empty(R * (U + 1), dtype=torch.bool, device=self.device)
        self.windows_seq(phys, U + 1, 0, out_windows, slot_elems=slot_elems, legal_out=legal, valid_out=mask)
        return legal, mask


def _topk_indices(keys, k):
    """Indices of the k largest keys (any order).  torch.topk over a replay of the reference's size (25 M positions, float64) is a
    2-ms radix select -- longer than the learner's step.  Exact shortcut: cut the keys into chunks; an element outside the k chunks
    with the largest maxima cannot beat the k-th of those maxima, so the k largest lie inside those k chunks (or in the ragged
    tail behind the last whole chunk): one row-maximum pass, a gather of k chunks, a top-k over k x chunk candidates
    (25 M positions: 1.9 -> 0.3 ms).  Small replays take torch.topk."""
    n = keys.numel()
    if n < (1 << 18) or n < 16 * k:
        return torch.topk(keys, k, sorted=False).indices
    chunk = min(4096, 1 << ((n // (4 * k)).bit_length() - 1))       # >= 4 k chunks, each a power of two long
    C = n // chunk
    rows = keys[:C * chunk].view(C, chunk)
    best = torch.topk(rows.amax(1), k, sorted=False).indices          # the k chunks that can hold winners
    cand = torch.cat((rows[best].reshape(-1), keys[C * chunk:]))      # ... and the tail (< chunk elements)
    inner = torch.topk(cand, k, sorted=False).indices
    in_rows = inner < k * chunk
    safe = torch.where(in_rows, inner, torch.zeros_like(inner))
    return torch.where(in_rows, best[safe // chunk] * chunk + safe % chunk, inner - k * chunk + C * chunk)


def policy_re_device(config, engine, windows, legal, mask, noises=None, generator=None, tie_seed=0, padded=False, roots=None):
    """reanalyze.prepare_policy_re (BatchWorker_GPU._prepare_policy_re, reanalyze_worker.py:307-371) with everything on the
    device and nothing read back: windows [B', stack * slot] in the engine's dtype, legal [B', A] uint8, mask [B'] bool ->
    policy targets [B', A] float32 (visit distribution over ALL children, zero rows where mask is False)."""
    from . import cytree
    from .mcts import MCTS
    B, A = windows.shape[0], config.action_space_size
    d = windows.device
    with torch.no_grad():
        _, logits, hidden = engine.initial(windows, padded=padded)
        if noises is None:
            alpha = torch.full((B, A), float(config.root_dirichlet_alpha), dtype=torch.float64, device=d)
            gm = torch._standard_gamma(alpha, generator=generator)
            noises = (gm / gm.sum(1, keepdim=True)).to(torch.float32)
        noises = noises * legal.to(torch.float32)                               # reanalyze_worker.py:344
        if roots is None:
            roots = cytree.Roots(B, A, config.num_simulations, device=d, tie_seed=tie_seed)
        roots.tie_seed = int(tie_seed)
        roots.prepare(config.root_exploration_fraction, noises, torch.zeros(B, device=d), logits, legal)
        MCTS(config).run_multi(roots, engine, hidden)
        dist = roots.distributions_tensor().to(torch.float64)
        policy = dist / dist.sum(1, keepdim=True)
        return torch.where(mask[:, None], policy, torch.zeros_like(policy)).to(torch.float32)
