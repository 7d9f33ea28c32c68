"""hanabizero_amd.selfplay -- the self-play actor of HanabiZero, one process per MI355X, device-resident.

What it replaces: ``DataWorker.run_multi`` (/root/reference/core/selfplay_worker.py:91-393).  Per lock-step the
reference does, per env and in Python: stack observations (:258-265), initial inference (:269-276), build a new
cytree.Roots + Dirichlet noise (:278-280), MCTS (:282), read distributions/values (:284-285), then a loop over envs:
select_action, env.step, store_search_stats, append, episode bookkeeping (:286-347) and, for finished games,
game_over + put + save_pools + reset (:215-256).  Here all of that is ONE stream of kernels over the whole batch --
optionally captured once into a hipGraph and replayed per lock-step -- with no host round trip inside a move:

    initial inference -> hz_tree_prepare -> (S-1) x [hz_tree_traverse_gather -> dynamics/prediction GEMMs ->
    hz_tree_backprop] -> read-outs -> hz_select_action -> trajectory append -> hz_env_step -> hz_env_observe ->
    finished-game flush (hz_rows_scatter into an outbox ring) -> hz_env_reset(mask) -> hz_env_observe -> stack update

Finished games leave the device as fixed-layout packed records (``drain()``), which ``unpack_record`` /
``GameHistory.from_packed`` turn back into reference-shaped histories on the replay side, and which
``hanabizero_amd.dist.gather_records`` moves to the replay owner's rank.

Reference behaviours kept: num_simulations-1 simulations; illegal root children can collect visits and are masked at
action selection; child_visits are the MASKED counts normalised by their sum; the terminal observation is stored;
rewards are raw score deltas (put()'s turn-reward reshape is applied by the consumer: game.reshape_turn_rewards).
Deliberate deviations: env i is seeded ``seed + global_env_id`` (the reference seeds every env of actor
rank 0 identically: selfplay_worker.py:102); Dirichlet noise and sampling uniforms come from a counter-based device
stream keyed by (seed, global env id, move number) (hz_actor_draw) instead of numpy's global generator; tie-breaks
from include/hz_tiebreak.h.
"""
import ctypes as C

import numpy as np
import torch

from . import cytree
from ._lib import ActorBufs, RowsJob, check, lib
from .hanabi_env import HanabiVecEnv
from .mcts import MCTS


def done_dtype_ok(env):
    """the bookkeeping kernels read the env's step outputs as (i32 reward, u8 done, i32 score, i32 status)"""
    return (env.reward.dtype, env.done.dtype, env.score.dtype, env.status.dtype) == \
        (torch.int32, torch.uint8, torch.int32, torch.int32)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SelfPlayActor:
    # predicted_lines="auto": mean of the trees' `path_len` after a search (hz_tree_get_path_len: the LAST simulation's path) at which
    # the lock-steps change kernels.  Measured at 4096 envs (`--net sharp:s`): policies whose searches end on paths
    # of 4.3-4.6 (random-init nets, s = 5, 10) lose 0.6-1.1 % to the predicted-line kernels, 5.65 (s = 20) gains 2.3 %, 7.9 (s = 40) 20 %.
    PATH_LINES_ON, PATH_LINES_OFF = 5.3, 5.0

    def __init__(self, config, engine, num_envs, rank=0, seed=0, device=None, use_graph=True, outbox_games=None,
                 deterministic=False, env_id_base=None, stream=None, fused_tail=True, predicted_lines="auto", root_noise=True,
                 tie_seed=None):
        """root_noise=False + deterministic=True: the evaluation protocol (core/test.py:93, 105: prepare_no_noise, first arg-max of the
        legal-masked visit counts) on the same lock-step; tie_seed: the tree's tie-break stream if not `seed`.
        predicted_lines: which persistent search kernels the lock-steps launch (include/hz_search.h::
        hz_search_set_predicted_lines): True = the ones whose descent walks predicted lines in trees that have grown deep (what a
        sharp policy needs), False = the plain ones (0.6-1 % faster while the trees stay shallow), "auto" = start plain and follow
        the mean length of the searches' last paths, looked at whenever finished games are drained (`PATH_LINES_ON` /
        `PATH_LINES_OFF`; a switch re-captures the lock-step's hipGraph).  Same records either way."""
        self.cfg, self.engine, self.N = config, engine, int(num_envs)
        self.predicted_lines = predicted_lines
        self._lines_on = predicted_lines is True
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        d, N = self.device, self.N
        # global id of this actor's env 0: keys the env seeds and the tie-break stream, so results do not depend on
        # how the envs are sharded over GPUs or over concurrent actors of one GPU
        self.env_id_base = rank * N if env_id_base is None else int(env_id_base)
        self.stream = stream  # optional side stream: several actors of one GPU overlap their (latency-bound) kernels
        self.deterministic = deterministic
        # everything after the search in two launches (include/hz_movetail.h) instead of one per phase; same bits (tests)
        self.fused_tail = bool(fused_tail)
        seeds = seed + self.env_id_base + np.arange(N)
        self.env = HanabiVecEnv(config.env_name, seeds, device=d, mdp=config.mdp)
        self.A, self.D = self.env.num_moves, self.env.obs_dim
        assert self.A == config.action_space_size and self.D == config.obs_dim
        self.S, self.stack, self.T = config.num_simulations, config.stacked_observations, config.max_moves
        self.W = self.env.packed_words
        self.root_noise = bool(root_noise)
        self.roots = cytree.Roots(N, self.A, self.S, device=d, tie_seed=seed if tie_seed is None else tie_seed, tree_id_base=self.env_id_base)
        self.roots.set_predicted_lines(self._lines_on)
        self.mcts = MCTS(config)
        dt = engine.dtype
        z = lambda *s, dtype: torch.zeros(s, dtype=dtype, device=d)
        # model input window and search state (static addresses: the whole step can be graph-captured).  Every slot of
        # the window is padded to a multiple of 8 elements (zero weights on the pad columns): 16-byte aligned rows for
        # the first GEMM and for the window shift
        self.Dp = engine.pad_observations(self.D, self.stack) if hasattr(engine, "pad_observations") else self.D
        self.stack_buf = z(N, self.stack, self.Dp, dtype=dt)
        self.newest = z(N, self.Dp, dtype=dt)
        self.legal = z(N, self.A, dtype=torch.uint8)
        self.pool = z(self.S, N, engine.H, dtype=dt)
        self.noise = z(N, self.A, dtype=torch.float32)
        self.uniform = z(N, dtype=torch.float64)
        self.zeros_n = z(N, dtype=torch.float32)
        self.action = z(N, dtype=torch.int32)
        self.entropy = z(N, dtype=torch.float64)
        self.ar = torch.arange(N, device=d)
        # per-env trajectory under construction
        T, A, W = self.T, self.A, self.W
        self.traj = dict(action=z(N, T, dtype=torch.int8), reward=z(N, T, dtype=torch.int8),
                         value=z(N, T, dtype=torch.float32), visits=z(N, T, A, dtype=torch.int16),
                         legal=z(N, T + 1, A, dtype=torch.uint8), obs=z(N, T + 1, W, dtype=torch.int32))
        self.traj_len = z(N, dtype=torch.int64)
        self.ent_sum = z(N, dtype=torch.float64)
        self.meta = z(N, 4, dtype=torch.int32)  # len, final score, global env id, visit-entropy sum (f32 bits)
        # outbox ring of finished games
        self.cap = int(outbox_games or max(4 * N, 1024))
        self.out = {k: torch.zeros((self.cap,) + v.shape[1:], dtype=v.dtype, device=d) for k, v in self.traj.items()}
        self.out_meta = z(self.cap, 4, dtype=torch.int32)
        self.out_count = z(2, dtype=torch.int64)   # games finished so far | their total moves (both cumulative)
        self.slot = z(N, dtype=torch.int32)
        self.tmp_packed = z(N, W, dtype=torch.int32)
        self.tmp_legal = z(N, A, dtype=torch.uint8)
        self.illegal_steps = z(1, dtype=torch.int64)
        self.finished = z(N, dtype=torch.int32)
        self.num_finished = z(1, dtype=torch.int32)
        self.counts = z(N, A, dtype=torch.int32)
        self.values = z(N, dtype=torch.float32)
        self._tail_scratch = z(2, dtype=torch.int64)
        # visit_softmax_temperature_fn's value in device memory: the fused tail reads it when it runs, so a captured lock-step
        # follows the schedule (set_trained_steps)
        self.temperature = torch.full((1,), float(config.visit_softmax_temperature_fn(0, 0)), dtype=torch.float32, device=d)
        assert done_dtype_ok(self.env)
        tr, o = self.traj, self.out
        self.bufs = ActorBufs(num_envs=N, num_actions=A, packed_words=W, max_moves=T, outbox_games=self.cap,
                              env_id_base=self.env_id_base, action=tr["action"].data_ptr(),
                              reward=tr["reward"].data_ptr(), value=tr["value"].data_ptr(),
                              visits=tr["visits"].data_ptr(), legal=tr["legal"].data_ptr(), obs=tr["obs"].data_ptr(),
                              traj_len=self.traj_len.data_ptr(), ent_sum=self.ent_sum.data_ptr(),
                              meta=self.meta.data_ptr(), out_action=o["action"].data_ptr(),
                              out_reward=o["reward"].data_ptr(), out_value=o["value"].data_ptr(),
                              out_visits=o["visits"].data_ptr(), out_legal=o["legal"].data_ptr(),
                              out_obs=o["obs"].data_ptr(), out_meta=self.out_meta.data_ptr(),
                              out_count=self.out_count.data_ptr(), slot=self.slot.data_ptr(),
                              finished=self.finished.data_ptr(), num_finished=self.num_finished.data_ptr(),
                              illegal_steps=self.illegal_steps.data_ptr())
        self.flush_job = RowsJob()  # hz_actor_flush as data: the masked reset of every lock-step carries it (hz_env_reset_rows)
        check(lib.hz_actor_flush_job(C.byref(self.bufs), C.byref(self.flush_job)), "hz_actor_flush_job")
        self.total_moves = 0
        self._drained = 0          # games / moves handed out by drains so far (host-side mirrors of out_count)
        self._moves_drained = 0
        self.drain_stream = None   # side stream of the asynchronous drain (drain_begin / drain_end), created on first use
        self._snap = None
        self._drawn = False  # the coming move's noise / uniforms are already in self.noise / self.uniform
        self._graph = None
        self.use_graph = use_graph
        self.noise_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.move_count = z(N, dtype=torch.int64)  # per-env draw counter of the noise stream
        self._start()

    # -- episode start for every env (selfplay_worker.py:118-138) ----------------------------------------
    def _start(self):
        self.env.reset()
        self.env.observe(out=self.newest, packed=self.tmp_packed, legal=self.legal)
        self.stack_buf.copy_(self.newest[:, None, :].expand(-1, self.stack, -1))
        self.traj["obs"][:, 0] = self.tmp_packed
        self.traj["legal"][:, 0] = self.legal
        self.traj_len.zero_()
        torch.cuda.synchronize(self.device)

    # -- randomness of one lock-step: root noise and sampling uniforms, drawn on the device (captured with the step) --
    def _draw(self):
        check(lib.hz_actor_draw(self.noise_seed, self.env_id_base, self.move_count.data_ptr(), self.N, self.A,
                                float(self.cfg.root_dirichlet_alpha), self.noise.data_ptr(), self.uniform.data_ptr(),
                                _stream()), "hz_actor_draw")

    # -- one lock-step, device only --------------------------------------------------------------------------
    def root_inference(self, state_out=None):
        """initial_inference of every env's current (slot-padded) observation window: (value, policy logits, state)."""
        return self.engine.initial(self.stack_buf.view(self.N, self.stack * self.Dp), state_out=state_out,
                                   padded=self.Dp != self.D)

    def _step_body(self, draw=True):
        """draw=True: the lock-step draws its own root noise / sampling uniforms -- the first one up front, every later
        one at the end of the step before (in the launch that moves the observation windows); draw=False: the caller has
        called _draw() for this move (tests)."""
        if draw and not self._drawn:
            self._draw()
        self._search_part()
        self._tail_part(draw)

    def _search_part(self):
        """Root inference, root preparation and the search of every env's current position (the part of a lock-step that reads
        the env and writes only the trees and the pool)."""
        cfg = self.cfg
        value0, logits0, hidden0 = self.root_inference(state_out=self.pool[0])
        if self.root_noise:
            self.roots.prepare(cfg.root_exploration_fraction, self.noise, self.zeros_n, logits0, self.legal)
        else:
            self.roots.prepare_no_noise(self.zeros_n, logits0, self.legal)
        self.mcts.run_multi(self.roots, self.engine, hidden0, pool=self.pool)

    def _tail_is_fused(self):
        return self.fused_tail and self.Dp * self.stack_buf.element_size() % 16 == 0

    def _tail_part(self, draw=True):
        """Everything after the search: read-out, action, env step, history, finished games, reset, next observation and window
        (and, draw=True, the next move's draws)."""
        cfg, N = self.cfg, self.N
        b, st = C.byref(self.bufs), _stream()
        if draw and self._tail_is_fused():
            # read-out, action, env step, history append | hand-over of finished games, reset, observation, window, next draws
            from .hanabi_env import _OBS_DTYPES
            env, es = self.env, self.stack_buf.element_size()
            check(lib.hz_actor_move_tail(self.roots._h, env._h, b, env.mdp, self.counts.data_ptr(), self.values.data_ptr(),
                                         self.legal.data_ptr(), self.uniform.data_ptr(), 1.0, self.temperature.data_ptr(),
                                         int(self.deterministic), self.action.data_ptr(), self.entropy.data_ptr(),
                                         env.reward.data_ptr(), env.done.data_ptr(), env.score.data_ptr(), env.status.data_ptr(),
                                         self.tmp_packed.data_ptr(), self.stack_buf.data_ptr(), self.stack_buf.stride(0) * es,
                                         self.stack, self.Dp * es, _OBS_DTYPES[self.stack_buf.dtype], self.noise_seed,
                                         self.move_count.data_ptr(), float(cfg.root_dirichlet_alpha), self.noise.data_ptr(),
                                         self._tail_scratch.data_ptr(), st), "hz_actor_move_tail")
            self._drawn = True
            return
        self.roots.root_stats_tensors(self.counts, self.values)
        check(lib.hz_actor_record_search(b, self.counts.data_ptr(), self.values.data_ptr(), self.legal.data_ptr(),
                                         self.uniform.data_ptr(), float(getattr(self, "_temperature_host", cfg.visit_softmax_temperature_fn(0, 0))),
                                         int(self.deterministic), self.action.data_ptr(), self.entropy.data_ptr(), st),
              "hz_actor_record_search")
        reward, done, score, status = self.env.step(self.action)
        # the observation after the move (terminal one included: selfplay_worker.py:308)
        self.env.observe_packed(self.tmp_packed, self.tmp_legal)
        check(lib.hz_actor_record_step(b, reward.data_ptr(), done.data_ptr(), score.data_ptr(), status.data_ptr(),
                                       self.tmp_packed.data_ptr(), self.tmp_legal.data_ptr(), st), "hz_actor_record_step")
        # finished games -> outbox ring (hz_actor_flush), in the launch that resets the finished envs
        # (selfplay_worker.py:230-240); then everybody's current observation
        # (tried: flush and the draws as parallel graph branches -- the cross-queue dependencies cost what the overlap saves)
        self.env.reset(done, rows=self.flush_job)
        self.env.observe(out=self.newest, packed=self.tmp_packed, legal=self.legal)
        # trajectory heads + stack window: shift for running games, refill for new ones (selfplay_worker.py:237, 326-327)
        es = self.newest.element_size()
        if draw:  # ... and the next move's draws in the same launch
            check(lib.hz_actor_begin_move_draw(b, done.data_ptr(), self.tmp_packed.data_ptr(), self.legal.data_ptr(),
                                               self.newest.data_ptr(), self.newest.stride(0) * es, self.stack_buf.data_ptr(),
                                               self.stack_buf.stride(0) * es, self.stack, self.Dp * es, self.noise_seed,
                                               self.move_count.data_ptr(), float(cfg.root_dirichlet_alpha),
                                               self.noise.data_ptr(), self.uniform.data_ptr(), st), "hz_actor_begin_move_draw")
            self._drawn = True
        else:
            check(lib.hz_actor_begin_move(b, done.data_ptr(), self.tmp_packed.data_ptr(), self.legal.data_ptr(),
                                          self.newest.data_ptr(), self.newest.stride(0) * es, self.stack_buf.data_ptr(),
                                          self.stack_buf.stride(0) * es, self.stack, self.Dp * es, st), "hz_actor_begin_move")
            self._drawn = False

    def set_trained_steps(self, trained_steps):
        """selfplay_worker.py:172-174: the visit-count temperature of the coming moves, from the learner's step counter
        (config.visit_softmax_temperature_fn; 1.0 throughout while change_temperature is off, as in both Hanabi configs).  In place:
        a captured lock-step picks it up at its next replay (fused tail; the launch-per-phase tail takes it at enqueue time)."""
        t = float(self.cfg.visit_softmax_temperature_fn(0, int(trained_steps)))
        with torch.cuda.stream(self._work_stream()):  # (ordered with the lock-steps already enqueued on the actor's own stream)
            self.temperature.fill_(t)
        if t != getattr(self, "_temperature_host", None) and self._graph is not None and not self._tail_is_fused():
            # (the launch-per-phase tail takes the temperature as a by-value argument: a captured graph has the old one baked in)
            self._retired_graph, self._graph = self._graph, None
        self._temperature_host = t

    def _capture(self):
        self.roots.set_params(self.cfg.pb_c_base, self.cfg.pb_c_init, self.cfg.discount, self.cfg.value_delta_max)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up outside capture (hipBLASLt workspaces, allocator)
            for _ in range(2):
                self._step_body()
                self.total_moves += self.N
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(self.device)
        self._retired_graph = None  # (nothing of a graph that _follow_path_lengths retired is in flight any more)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):  # other threads (e.g. the RCCL watchdog) may call HIP meanwhile
            self._step_body()
        self._graph = g

    def step(self):
        """Advance every env by one move (enqueue only; on `self.stream` when one was given)."""
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._step()
        else:
            self._step()

    def _step(self):
        if self.use_graph and self._graph is None:
            self._capture()
        if self.use_graph:
            self._graph.replay()
        else:
            self._step_body()
        self.total_moves += self.N

    # -- finished games ------------------------------------------------------------------------------------------
    def _work_stream(self):
        return self.stream if self.stream is not None else torch.cuda.current_stream(self.device)

    def drain(self):
        """Packed records of the games finished since the last drain (synchronises).  Returns a dict of numpy arrays
        with a leading games axis, or None.  Must be called at least once per `cap` finished games."""
        self._work_stream().synchronize()
        if self.predicted_lines == "auto":
            self._follow_path_lengths(float(self.roots.path_len_tensor().float().mean()))
        count = int(self.out_count[0].item())
        n = count - self._drained
        if n <= 0:
            return None
        if n > self.cap:
            raise RuntimeError("outbox overflow: %d games finished since the last drain, capacity %d" % (n, self.cap))
        idx = torch.arange(self._drained, count, device=self.device) % self.cap
        meta = self.out_meta.index_select(0, idx)
        tmax = int(meta[:, 0].max().item())  # longest finished game: only that many time slots leave the device
        rec = {}
        for k, v in self.out.items():
            t = tmax + 1 if k in ("legal", "obs") else tmax
            rec[k] = v.index_select(0, idx)[:, :t].contiguous().cpu().numpy()
        rec["meta"] = meta.cpu().numpy()
        self._drained = count
        self._moves_drained += int(rec["meta"][:, 0].sum())
        return rec

    def drain_begin(self):
        """First half of a drain that never stalls the lock-steps: snapshot (finished games, their total moves) as of
        the work enqueued so far.  Non-blocking.  The snapshot itself is taken ON THE WORK STREAM -- a 16-byte device copy
        between two lock-steps, where the pair is consistent and every counted game's rows are in the outbox (inside a
        lock-step the two counters are written by separate stores and the rows only by its later reset launch) -- and
        `drain_stream` waits for that copy's event, for nothing enqueued later, before it takes the snapshot to pinned host
        memory together with the inference kernels' give-up counters (include/hz_mlp.h::hz_mlp_poll_giveups_async)."""
        if self.drain_stream is None:
            self.drain_stream = torch.cuda.Stream(device=self.device)
            self._count_host = torch.zeros(2, dtype=torch.int64).pin_memory()
            self._count_snap = torch.zeros(2, dtype=torch.int64, device=self.device)
            self._giveups_host = torch.zeros(2, dtype=torch.int32).pin_memory()
            self._plen_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._plen_snap = torch.zeros(1, dtype=torch.float32, device=self.device)
            from ._lib import poll_giveups
            self._giveups_seen = poll_giveups()  # (what the process had before this actor's first drain -- e.g. a test's broken table -- is not this actor's)
        assert self._snap is None, "drain_begin: the previous snapshot has not been consumed (drain_end)"
        ws = self._work_stream()
        with torch.cuda.stream(ws):
            self._count_snap.copy_(self.out_count)
            if self.predicted_lines == "auto":  # (the last search's path lengths: what decides about the search kernels, drain_end)
                self._plen_snap.copy_(self.roots.path_len_tensor().float().mean())
            taken = torch.cuda.Event()
            taken.record(ws)
        self.drain_stream.wait_event(taken)
        with torch.cuda.stream(self.drain_stream):
            self._count_host.copy_(self._count_snap, non_blocking=True)
            self._plen_host.copy_(self._plen_snap, non_blocking=True)
            check(lib.hz_mlp_poll_giveups_async(self._giveups_host.data_ptr(), _stream()), "hz_mlp_poll_giveups_async")
            self._snap = torch.cuda.Event()
            self._snap.record(self.drain_stream)

    def drain_end(self):
        """Second half: the games finished up to the snapshot of drain_begin as ONE byte buffer that stays on the device:
        (uint8 tensor, n games, their total moves) or None.  Blocks the host until the snapshot has been taken (i.e. until
        the lock-steps enqueued BEFORE drain_begin are done), not for lock-steps enqueued since; the packing runs on
        `drain_stream` beside them.  The buffer belongs to `drain_stream`: consume it under
        ``with torch.cuda.stream(actor.drain_stream)`` (what bench.py does with dist.gather_packed) or make your stream
        wait for it.  `packed_layout(n, moves, A, W)` describes the buffer; `unpack_packed` views one on the host."""
        assert self._snap is not None, "drain_end without drain_begin"
        self._snap.synchronize()
        self._snap = None
        count, moves_total = (int(x) for x in self._count_host.tolist())
        self._follow_path_lengths(float(self._plen_host[0]))
        giveups = int(self._giveups_host[0]) + int(self._giveups_host[1])
        if giveups != self._giveups_seen:  # a wait on an arrival counter timed out: search results since the last drain are not to be trusted
            seen, self._giveups_seen = self._giveups_seen, giveups
            raise RuntimeError("the fused inference gave up %d wait(s) on its arrival counters since the last drain "
                               "(a job table that breaks include/hz_mlp.h's contract, or a stalled wave): the games of this "
                               "interval were searched with inputs that may not have been there" % (giveups - seen))
        n, moves = count - self._drained, moves_total - self._moves_drained
        if n <= 0:
            return None
        if n > self.cap:
            raise RuntimeError("outbox overflow: %d games finished since the last drain, capacity %d" % (n, self.cap))
        total = int(lib.hz_actor_packed_bytes(n, moves, self.A, self.W, None))
        with torch.cuda.stream(self.drain_stream):
            buf = torch.empty(total, dtype=torch.uint8, device=self.device)
            starts = torch.empty(n + 1, dtype=torch.int32, device=self.device)
            check(lib.hz_actor_pack(C.byref(self.bufs), self._drained, n, moves, starts.data_ptr(), buf.data_ptr(), total,
                                    _stream()), "hz_actor_pack")
        self._drained, self._moves_drained = count, moves_total
        return buf, n, moves

    def _follow_path_lengths(self, mean_nodes):
        """predicted_lines="auto": change the search kernels when the searches' last paths have grown long / short enough (the
        next lock-step captures its hipGraph again: two eager moves and a capture, ~10 ms, rare)."""
        if self.predicted_lines != "auto" or mean_nodes <= 0.0 or self.A > 20:  # (the kernels walk predicted lines for A <= 20 only)
            return
        hold = getattr(self, "_lines_hold", 0)
        self._lines_hold = max(0, hold - 1)
        want = mean_nodes >= self.PATH_LINES_ON if not self._lines_on else mean_nodes > self.PATH_LINES_OFF
        if want != self._lines_on and hold == 0:
            self._lines_hold = 4  # (a change costs a capture: the next four drains' readings are not acted on)
            self._lines_on = want
            self.roots.set_predicted_lines(want)
            # (replays of the old graph may still be in flight behind this drain: it stays alive until the next capture has
            # synchronised the device)
            self._retired_graph, self._graph = self._graph, None

    def drain_packed(self):
        """drain_begin + drain_end in one blocking call: the games finished by the work enqueued so far, usable on the
        caller's current stream.  (Two kernels and one 16-byte read-back; `hanabizero_amd.dist.gather_packed` moves such
        buffers to the replay owner GPU-to-GPU, no host copy on the sending ranks.)"""
        self.drain_begin()
        got = self.drain_end()
        if got is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.drain_stream)
            got[0].record_stream(cur)
        return got


class ActorGroup:
    """Several SelfPlayActors of one GPU stepped by ONE hipGraph whose branches (one per actor, forked onto side streams
    inside the capture) have no edges between them, so the runtime may overlap one actor's latency-bound tree kernels
    with another's GEMMs.  Results are those of the actors run one after the other (disjoint state)."""

    def __init__(self, actors):
        self.actors = list(actors)
        self.device = self.actors[0].device
        self._graph = None

    def set_trained_steps(self, trained_steps):
        """SelfPlayActor.set_trained_steps for every actor of the group."""
        for a in self.actors:
            a.set_trained_steps(trained_steps)

    def _capture(self):
        for a in self.actors:
            a.roots.set_params(a.cfg.pb_c_base, a.cfg.pb_c_init, a.cfg.discount, a.cfg.value_delta_max)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for a in self.actors:
                    a._step_body()
                    a.total_moves += a.N
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(self.device)
        self._retired_graph = None
        branches = [torch.cuda.Stream(device=self.device) for _ in self.actors]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):  # other threads (e.g. the RCCL watchdog) may call HIP meanwhile
            root = torch.cuda.current_stream()
            for a, b in zip(self.actors, branches):
                b.wait_stream(root)
                with torch.cuda.stream(b):
                    a._step_body()
            for b in branches:
                root.wait_stream(b)
        self._graph = g

    def step(self):
        if any(a._lines_on != on for a, on in zip(self.actors, getattr(self, "_lines_captured", ()))):
            # (an actor's drain has changed its search kernels: SelfPlayActor._follow_path_lengths; the old graph lives until
            # _capture has synchronised the device)
            self._retired_graph, self._graph = self._graph, None
        if self._graph is None:
            self._capture()
            self._lines_captured = [a._lines_on for a in self.actors]
        for a in self.actors:
            a.total_moves += a.N
        self._graph.replay()


_PACKED_FIELDS = ("meta", "action", "reward", "value", "visits", "legal", "obs")  # order inside a packed buffer


def packed_layout(n, moves, A, W):
    """[(field, shape, numpy dtype, byte offset)], total bytes of the packed form of n games with `moves` moves in total
    (SelfPlayActor.drain_packed, include/hz_selfplay.h hz_actor_pack): the games lie back to back in every section --
    game j at rows [start_j, start_j + len_j) of the per-move sections and [start_j + j, start_j + j + len_j + 1) of
    legal / obs, start_j = len_0 + .. + len_(j-1), len_j = meta[j][0] -- and every section starts on a 16-byte boundary."""
    shapes = dict(meta=((n, 4), np.int32), action=((moves,), np.int8), reward=((moves,), np.int8),
                  value=((moves,), np.float32), visits=((moves, A), np.int16), legal=((moves + n, A), np.uint8),
                  obs=((moves + n, W), np.int32))
    out, off = [], 0
    for k in _PACKED_FIELDS:
        shp, dt = shapes[k]
        out.append((k, shp, dt, off))
        off += (int(np.prod(shp)) * np.dtype(dt).itemsize + 15) // 16 * 16
    return out, off


def unpack_packed(buf, n, moves, A, W):
    """Zero-copy views of a packed byte buffer (numpy uint8, host): the seven sections plus "start" [n] = the games'
    first rows.  `unpack_record(rec, i)` cuts game i out of it."""
    layout, total = packed_layout(n, moves, A, W)
    assert buf.nbytes >= total
    rec = {k: buf[off:off + int(np.prod(shp)) * np.dtype(dt).itemsize].view(dt).reshape(shp) for k, shp, dt, off in layout}
    lens = rec["meta"][:, 0].astype(np.int64)
    assert int(lens.sum()) == moves, "packed buffer: the games' lengths add up to %d, not %d" % (int(lens.sum()), moves)
    rec["start"] = np.cumsum(lens) - lens
    return rec


def pack_records(rec, A, W):
    """Host-side packer: a drain() dict (arrays padded to the longest game) -> (uint8 numpy buffer, n, moves) in the format
    of drain_packed (tests, and actors that drained with drain())."""
    n = int(rec["meta"].shape[0])
    lens = rec["meta"][:, 0].astype(np.int64)
    moves = int(lens.sum())
    layout, total = packed_layout(n, moves, A, W)
    buf = np.zeros(total, np.uint8)
    views = {k: buf[off:off + int(np.prod(shp)) * np.dtype(dt).itemsize].view(dt).reshape(shp) for k, shp, dt, off in layout}
    views["meta"][:] = rec["meta"]
    s = 0
    for i in range(n):
        T = int(lens[i])
        for k in ("action", "reward", "value", "visits"):
            views[k][s:s + T] = rec[k][i, :T]
        for k in ("legal", "obs"):
            views[k][s + i:s + i + T + 1] = rec[k][i, :T + 1]
        s += T
    return buf, n, moves


def unpack_record(rec, i):
    """Game i of a drained batch (drain(): padded arrays; unpack_packed(): ragged sections) -> the dict
    GameHistory.from_packed takes."""
    meta = rec["meta"][i]
    T = int(meta[0])
    if "start" in rec:
        s = int(rec["start"][i])
        cut = {k: rec[k][s:s + T] for k in ("action", "reward", "value", "visits")}
        cut.update({k: rec[k][s + i:s + i + T + 1] for k in ("legal", "obs")})
    else:
        cut = {k: rec[k][i, :T] for k in ("action", "reward", "value", "visits")}
        cut.update({k: rec[k][i, :T + 1] for k in ("legal", "obs")})
    return dict(len=T, score=int(meta[1]), env_id=int(meta[2]),
                visit_entropy_sum=float(np.array([meta[3]], np.int32).view(np.float32)[0]),
                action=cut["action"], reward=cut["reward"], value=cut["value"], visits=cut["visits"], legal=cut["legal"],
                obs_bits=cut["obs"].view(np.uint32))


def record_nbytes(rec):
    return int(sum(v.nbytes for v in rec.values()))
