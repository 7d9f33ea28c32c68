"""hanabizero_amd.learner -- the learner step of HanabiZero on PyTorch-ROCm.

What it replaces: ``update_weights`` (/root/reference/core/train.py:59-314) and the optimiser / learning-rate plumbing
of ``_train`` (train.py:317-400, ``adjust_lr`` :32-51).  Same losses, same targets, same gradient shaping:

  * targets: ``scalar_transform`` (core/config.py:192-202) then the two-hot ``phi`` over the integer support
    (config.py:240-253);
  * losses: cross-entropy of the categorical value / reward heads against those two-hots and of the policy logits against
    the MCTS visit distributions (config/hanabi_control/__init__.py:119-123; train.py:145-168), summed over the initial
    step and ``num_unroll_steps`` recurrent steps, weighted by the prioritised-replay importance weights;
  * gradient shaping: the hidden state's gradient is halved at every unroll step (train.py:169) and the total loss's by
    ``1 / num_unroll_steps`` (train.py:222-229); ``clip_grad_norm_(max_grad_norm)`` (train.py:241);
  * priorities: ``(1 - r) * |value - target| + r * mean_k |reward_k - target_k|`` + eps (train.py:141-143, 250-252).

The reference runs the forward passes under fp16 ``autocast`` with a ``GradScaler`` (train.py:125, 153, 225-245).  On
MI355X the step runs under **bf16** autocast (hipBLASLt GEMMs on the matrix cores; no loss scaling needed with bf16's fp32
exponent range), or in plain fp32 with ``amp=None`` -- the mode the CPU tests pin the arithmetic in.  The consistency loss
is asserted off in the reference for Hanabi (train.py:158-160: "will not run here") and is not built.

Ray is gone: instead of ``replay_buffer.update_priorities.remote(...)`` the new priorities are returned to the caller.
The nets are the state_dict-compatible modules of ``hanabizero_amd.model``, so a learner started from a reference
checkpoint continues it, and ``hanabizero_amd.dist.broadcast_weights`` ships ``model.get_weights()`` to the actors.
"""
import numpy as np
import torch


# ---- targets (core/config.py) ------------------------------------------------------------------------------
def scalar_transform(x, delta=1.0, epsilon=0.001):
    """h(x) = sign(x) (sqrt(|x| + 1) - 1) + eps x   (core/config.py:192-202, with the reference's support delta)."""
    sign = torch.ones_like(x)
    sign[x < 0] = -1.0
    return sign * (torch.sqrt(torch.abs(x / delta) + 1) - 1) + epsilon * x / delta


def phi(x, support_min, support_max, support_size, delta=1.0):
    """Two-hot encoding of transformed scalars [B, T] over the integer support (core/config.py:240-253): mass
    ``x - floor(x)`` on ceil(x) and the rest on floor(x) (an integer x puts everything on itself: the second scatter
    overwrites the first, exactly as in the reference)."""
    x = x.clamp(support_min, support_max)
    x_low, x_high = x.floor(), x.ceil()
    p_high = x - x_low
    p_low = 1 - p_high
    target = torch.zeros(x.shape[0], x.shape[1], support_size, device=x.device, dtype=x.dtype)
    hi, lo = x_high - support_min / delta, x_low - support_min / delta
    target.scatter_(2, hi.long().unsqueeze(-1), p_high.unsqueeze(-1))
    target.scatter_(2, lo.long().unsqueeze(-1), p_low.unsqueeze(-1))
    return target


def scalar_loss(prediction, target):
    """config/hanabi_control/__init__.py:119-123 (value and reward alike)."""
    return -(torch.log_softmax(prediction, dim=1) * target).sum(1)


# ---- optimiser and schedule (core/train.py:32-51, 327-328) -------------------------------------------------------
def make_optimizer(model, config, capturable=False):
    """capturable: the fused implementation with the learning rate in a device tensor, so that a step can be captured
    into a hipGraph (GraphedUpdate) and adjust_lr still takes effect."""
    if capturable:
        dev = next(model.parameters()).device
        return torch.optim.SGD(model.parameters(), lr=torch.tensor(float(config.lr_init), device=dev), momentum=config.momentum,
                               weight_decay=config.weight_decay, fused=True)
    return torch.optim.SGD(model.parameters(), lr=config.lr_init, momentum=config.momentum,
                           weight_decay=config.weight_decay)


def adjust_lr(config, optimizer, step_count):
    """Linear warm-up over lr_warm_step steps, then step decay floored at 1e-4 (train.py:32-51, lr_type 'step')."""
    if step_count < config.lr_warm_step:
        lr = config.lr_init * step_count / config.lr_warm_step
    else:
        lr = config.lr_init * config.lr_decay_rate ** ((step_count - config.lr_warm_step) // config.lr_decay_steps)
        lr = lr if lr >= 0.0001 else 0.0001
    for group in optimizer.param_groups:
        if isinstance(group["lr"], torch.Tensor):
            group["lr"].fill_(lr)  # (in place: a captured step reads the tensor)
        else:
            group["lr"] = lr
    return lr


# ---- batches (core/reanalyze_worker.py, the parts that do not search) ---------------------------------------------
_SCRATCH = {}


def _scratch(name, shape, dtype):
    """A host buffer that lives across calls (contents undefined on return)."""
    key = (name, tuple(shape), np.dtype(dtype).str)
    buf = _SCRATCH.get(key)
    if buf is None:
        for k in [k for k in _SCRATCH if k[0] == name]:
            del _SCRATCH[k]
        buf = _SCRATCH[key] = np.empty(shape, dtype)
    return buf


def make_batch(games, positions, config, value_fn, weights=None, rng=None, policy_re=None, obs_dtype=np.float32):
    """A learner batch in the reference's layout from finished ``GameHistory`` objects and sampled positions:
    inputs as BatchWorker_CPU.make_batch assembles them (reanalyze_worker.py:148-168: stacked observations padded with
    the last frame, actions padded with random ones past the end, mask), value / reward targets as
    BatchWorker_GPU._prepare_reward_value (:249-304: td_steps-step return bootstrapped from ``value_fn`` of the
    observation td_steps ahead, zero past the end), policy targets from the stored search statistics as
    _prepare_policy_non_re (:374-399: child visits, zeros past the end).  ``value_fn(obs [M, stack * D] float32 numpy)
    -> [M] values`` is the target model (e.g. ``lambda o: engine.initial(torch.from_numpy(o).cuda())[0].cpu().numpy()``).
    policy_re: [R, U + 1, A] policy targets re-searched with the target model for the FIRST R positions of the batch
    (hanabizero_amd.reanalyze.prepare_policy_re over reanalyze.policy_re_context(config, games[:R], positions[:R])); they
    replace those rows' stored search statistics, as _prepare_target_gpu concatenates [reanalyzed | stored] (:412-419).
    obs_dtype: element type of the observation arrays this function fills (the batch's `obs_batch` and the bootstrap windows
    handed to value_fn).  float32 is the reference's; np.uint8 keeps the frames as they are stored (0 / 1 bytes: a quarter of
    the host traffic and of the PCIe bytes -- 12 MB instead of 47 MB per Hanabi-Full-5p batch); update_weights /
    GraphedUpdate and the engines convert on the device."""
    rng = rng or np.random
    U, td, stack, A, g = config.num_unroll_steps, config.td_steps, config.stacked_observations, config.action_space_size, config.discount
    B = len(games)
    D = config.obs_shape // stack
    # Outputs are written in place, per sample a handful of array operations (the straightforward version -- a Python loop
    # per unroll step and per reward term, `tests/test_learner.py::make_batch_spec` -- was 9 ms of a 23 ms learner step at batch
    # 256).  Same values bit for bit: the reward terms are added in the reference's order, i = 0 .. td - 1, one vector step each.
    obs_batch = np.empty((B, stack + U, D), obs_dtype)
    # (the bootstrap windows go to value_fn and nowhere else: one buffer, kept across calls -- a fresh 8 MB array per batch costs
    # more in page faults than everything else here -- rows past the end of a game zeroed where there are any: zero_obs)
    value_obs = _scratch("value_obs", (B * (U + 1), config.obs_shape), obs_dtype)
    value_mask = np.zeros(B * (U + 1), np.float64)
    actions = np.empty((B, U), np.int64)
    mask = np.zeros((B, U), np.float32)
    rew_win = np.zeros((B, U + 1, td), np.float64)    # rewards[cur + i], zero past the end of the game
    lens = np.empty(B, np.int64)
    per_sample = []
    windows = np.lib.stride_tricks.sliding_window_view
    for b, (game, pos) in enumerate(zip(games, positions)):
        acts_all, rewards, traj_len = game.actions, np.asarray(game.rewards, np.float64), len(game)
        n = min(U, traj_len - pos)
        actions[b, :n] = acts_all[pos:pos + n]
        mask[b, :n] = 1.0
        for j in range(n, U):
            actions[b, j] = int(rng.randint(0, A))   # (random actions past the end, drawn in the reference's order)
        obs_batch[b] = game.obs(pos, extra_len=U, padding=True)
        # bootstrap observations (:204-222): the windows td steps ahead of every unroll position that still has one
        nv = min(U + 1, max(0, traj_len - td - pos))
        k0 = b * (U + 1)
        if nv > 0:
            frames = np.asarray(game.obs(pos + td, nv - 1))
            for j in range(nv):
                value_obs[k0 + j].reshape(stack, D)[:] = frames[j:j + stack]
            value_mask[k0:k0 + nv] = 1.0
        if nv < U + 1:
            value_obs[k0 + nv:k0 + U + 1] = 0
        seg = rewards[pos:pos + U + td]
        if len(seg) < U + td:
            seg = np.concatenate((seg, np.zeros(U + td - len(seg))))
        rew_win[b] = windows(seg, td)
        lens[b] = traj_len
        per_sample.append((game, pos, rewards))
    values = np.asarray(value_fn(value_obs), dtype=np.float64).reshape(-1) * (g ** td) * value_mask
    v = values.reshape(B, U + 1).copy()
    for i in range(td):
        v += rew_win[:, :, i] * g ** i   # (a term past the end of the game is + 0.0)
    target_value = np.zeros((B, U + 1), np.float32)
    target_reward = np.zeros((B, U + 1), np.float32)
    target_policy = np.zeros((B, U + 1, A), np.float32)
    for b, (game, pos, rewards) in enumerate(per_sample):
        n = min(U + 1, int(lens[b]) - pos)
        if n > 0:
            target_value[b, :n] = v[b, :n]
            target_reward[b, :n] = rewards[pos:pos + n]
            target_policy[b, :n] = game.child_visits[pos:pos + n]
    if policy_re is not None and len(policy_re):
        target_policy[:len(policy_re)] = policy_re
    w = np.ones(B, np.float32) if weights is None else np.asarray(weights, np.float32)
    inputs = (obs_batch, actions, mask, np.arange(B), w, np.zeros(B))
    return inputs, (target_reward[:, :U + 1], target_value, target_policy)


# ---- one learner step ------------------------------------------------------------------------------------------
def _t(a, device, dtype=torch.float32):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))  # (a strided host view crosses PCIe row by row)
    if t.dtype == torch.uint8 and t.device != torch.device(device):
        return t.to(device).to(dtype)  # (0 / 1 frames travel as bytes and are widened on the device)
    return t.to(device=device, dtype=dtype)


def _support_to_scalar(logits, support):
    """inverse_value / inverse_reward_transform (core/config.py:210-232) of head logits [B, 2 s + 1] for the PRIORITIES (no
    gradient): on the GPU the HIP kernel the search uses (include/hz_tree.h hz_support_to_scalar: one launch, any of the three
    element formats) instead of ~15 elementwise launches per inference."""
    logits = logits.detach()
    if not logits.is_cuda or logits.dim() != 2 or logits.stride(1) != 1:
        from .model import inverse_scalar_transform
        return inverse_scalar_transform(logits.float(), support.min, support.max).reshape(-1)
    import ctypes as C
    from ._lib import check, lib
    out = torch.empty(logits.shape[0], dtype=torch.float32, device=logits.device)
    dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[logits.dtype]
    check(lib.hz_support_to_scalar(logits.data_ptr(), logits.stride(0), support.size, support.min, dt, out.data_ptr(), logits.shape[0],
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hz_support_to_scalar")
    return out


def compute_losses(model, config, obs_batch, action_batch, target_reward, target_value, target_policy, weights,
                   amp=torch.bfloat16):
    """The forward part of update_weights (train.py:114-222).  Tensors on the model's device:
    obs_batch [B, stack * D] (the first stacked window of the batch), action_batch [B, U] long, target_reward [B, U],
    target_value / target_policy [B, U + 1] / [B, U + 1, A], weights [B].
    Returns (weighted_loss, dict of per-sample losses and priorities' ingredients)."""
    if getattr(model, "fused_heads", False) and obs_batch.is_cuda and amp is not None:
        return model.compute_losses(config, obs_batch, action_batch, target_reward, target_value, target_policy, weights)
    U = config.num_unroll_steps
    B = obs_batch.shape[0]
    dev = obs_batch.device
    vs, rs = config.value_support, config.reward_support
    target_reward_phi = phi(scalar_transform(target_reward, vs.delta), rs.min, rs.max, rs.size, vs.delta)
    target_value_phi = phi(scalar_transform(target_value, vs.delta), vs.min, vs.max, vs.size, vs.delta)

    def cast():
        return torch.autocast(device_type=dev.type, dtype=amp) if amp is not None else torch.autocast(dev.type, enabled=False)

    with cast():
        value, _, policy_logits, hidden_state = model.initial_inference(obs_batch.reshape(B, -1))
    scaled_value = _support_to_scalar(value, vs)
    value_priority = (scaled_value.reshape(B) - target_value[:, 0]).abs().detach()
    value_loss = scalar_loss(value.float(), target_value_phi[:, 0])
    policy_loss = -(torch.log_softmax(policy_logits.float(), dim=1) * target_policy[:, 0]).sum(1)
    reward_loss = torch.zeros(B, device=dev)
    reward_priority = []
    with cast():
        for k in range(U):
            value, reward, policy_logits, hidden_state = model.recurrent_inference(hidden_state, action_batch[:, k:k + 1])
            policy_loss = policy_loss - (torch.log_softmax(policy_logits.float(), dim=1) * target_policy[:, k + 1]).sum(1)
            value_loss = value_loss + scalar_loss(value.float(), target_value_phi[:, k + 1])
            reward_loss = reward_loss + scalar_loss(reward.float(), target_reward_phi[:, k])
            if hidden_state.requires_grad:
                hidden_state.register_hook(lambda grad: grad * 0.5)  # train.py:169
            scaled_reward = _support_to_scalar(reward, rs)
            reward_priority.append((scaled_reward.reshape(B) - target_reward[:, k]).abs())
    loss = config.policy_loss_coeff * policy_loss + config.value_loss_coeff * value_loss + config.reward_loss_coeff * reward_loss
    weighted_loss = (weights * loss).mean()
    return weighted_loss, dict(loss=loss, policy_loss=policy_loss, value_loss=value_loss, reward_loss=reward_loss,
                               value_priority=value_priority, reward_priority=torch.stack(reward_priority).mean(0))


def update_weights(model, batch, optimizer, config, amp=torch.bfloat16):
    """One learner step on a batch in the reference's layout (train.py:59-70):
        batch = ((obs_batch_ori [B, stack + U, D], action_batch [B, U], mask_batch [B, U], indices, weights [B], make_time),
                 (target_reward [B, U], target_value [B, U + 1], target_policy [B, U + 1, A]))
    numpy arrays or tensors.  Returns (loss_data, new_priority [B] numpy) with loss_data = (total, weighted, mean loss,
    0, mean policy, mean reward, mean value, 0.0) as the reference logs it (train.py:255-256)."""
    (obs_batch_ori, action_batch, mask_batch, indices, weights, make_time), (target_reward, target_value, target_policy) = batch
    target_reward = np.asarray(target_reward)[:, :config.num_unroll_steps] if not isinstance(target_reward, torch.Tensor) \
        else target_reward[:, :config.num_unroll_steps]
    dev = next(model.parameters()).device
    # non-image branch of train.py:71-74: image_channel = 1, the first `stack` observations are the model input
    obs_batch = _t(obs_batch_ori[:, 0:config.stacked_observations, :], dev)
    action_batch = _t(action_batch, dev, torch.long)
    target_reward, target_value = _t(target_reward, dev), _t(target_value, dev)
    target_policy, weights = _t(target_policy, dev), _t(weights, dev)
    B = obs_batch.shape[0]
    assert B == target_reward.shape[0] and action_batch.shape[1] == config.num_unroll_steps

    model.train()
    weighted_loss, parts = compute_losses(model, config, obs_batch, action_batch, target_reward, target_value,
                                          target_policy, weights, amp=amp)
    total_loss = weighted_loss
    gradient_scale = 1.0 / config.num_unroll_steps
    total_loss.register_hook(lambda grad: grad * gradient_scale)  # train.py:222-229
    fused = hasattr(model, "refresh")  # (hanabizero_amd.fused_train.FusedTrainNet: its blocks accumulate into existing .grad tensors)
    optimizer.zero_grad(set_to_none=not fused)
    total_loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), config.max_grad_norm)
    optimizer.step()
    if fused:
        model.refresh()
        model.count_batches()
    r = config.priority_reward_ratio
    new_priority = (1 - r) * (parts["value_priority"] + config.prioritized_replay_eps) + r * parts["reward_priority"]
    f = lambda t: float(t.detach())
    loss_data = (f(total_loss), f(weighted_loss), f(parts["loss"].mean()), 0, f(parts["policy_loss"].mean()),
                 f(parts["reward_loss"].mean()), f(parts["value_loss"].mean()), 0.0)
    return loss_data, new_priority.detach().cpu().numpy()


class GraphedUpdate:
    """update_weights with the whole step -- forward through the unrolled model, backward, gradient clipping, SGD -- captured
    once into a hipGraph and replayed per batch: eager PyTorch spends the step launching ~1.5 k small kernels.
    Same arithmetic as update_weights (it captures compute_losses and the same optimiser calls); the optimiser must come
    from make_optimizer(model, config, capturable=True).  Batches must have the captured batch size.

        step = GraphedUpdate(model, optimizer, config, batch_size)
        loss_data, new_priority = step(batch)        # same returns as update_weights"""

    def __init__(self, model, optimizer, config, batch_size, amp=torch.bfloat16):
        self.model, self.optimizer, self.config, self.amp = model, optimizer, config, amp
        dev = next(model.parameters()).device
        B, U, A, stack = int(batch_size), config.num_unroll_steps, config.action_space_size, config.stacked_observations
        D = config.obs_shape // stack
        z = lambda *shape, dtype=torch.float32: torch.zeros(shape, dtype=dtype, device=dev)
        self.obs, self.action = z(B, stack, D), z(B, U, dtype=torch.long)
        self.target_reward, self.target_value, self.target_policy = z(B, U), z(B, U + 1), z(B, U + 1, A)
        self.target_policy[:] = 1.0 / A
        self.weights = z(B) + 1.0
        self.out = z(7)
        self.priority = z(B)
        self._graph = None

    def _body(self, io=None):
        """io: an object with the same input tensors (obs .. weights) and a `priority` output as this one (a batch slot of
        LearnerPipeline): the step reads / writes those instead."""
        cfg = self.config
        io = self if io is None else io
        weighted_loss, parts = compute_losses(self.model, cfg, io.obs, io.action, io.target_reward, io.target_value,
                                              io.target_policy, io.weights, amp=self.amp)
        total_loss = weighted_loss
        gradient_scale = 1.0 / cfg.num_unroll_steps
        total_loss.register_hook(lambda grad: grad * gradient_scale)
        self.optimizer.zero_grad(set_to_none=False)
        total_loss.backward()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), cfg.max_grad_norm, foreach=True)
        self.optimizer.step()
        if hasattr(self.model, "refresh"):  # (fused_train.FusedTrainNet: the 16-bit weight copies follow the step, inside the graph)
            self.model.refresh()
            self.model.count_batches()
        r = cfg.priority_reward_ratio
        io.priority.copy_((1 - r) * (parts["value_priority"] + cfg.prioritized_replay_eps) + r * parts["reward_priority"])
        self.out.copy_(torch.stack([total_loss.detach(), weighted_loss.detach(), parts["loss"].detach().mean(),
                                    parts["policy_loss"].detach().mean(), parts["reward_loss"].detach().mean(),
                                    parts["value_loss"].detach().mean(), total_loss.detach() * 0]))

    def _snapshot(self):
        """What a step changes (weights, BatchNorm statistics, momentum), so that steps taken for warm-up or timing can be undone."""
        mom = [None if st.get("momentum_buffer") is None else st["momentum_buffer"].detach().clone() for st in self.optimizer.state.values()]
        return [p.detach().clone() for p in self.model.parameters()], [b.detach().clone() for b in self.model.buffers()], mom

    def _restore(self, snap):
        saved, saved_buf, mom = snap
        with torch.no_grad():
            for p, q in zip(self.model.parameters(), saved):
                p.copy_(q)
            for b, q in zip(self.model.buffers(), saved_buf):
                b.copy_(q)
            states = list(self.optimizer.state.values())
            for i, st in enumerate(states):
                if st.get("momentum_buffer") is not None:
                    if i < len(mom) and mom[i] is not None:
                        st["momentum_buffer"].copy_(mom[i])
                    else:
                        st["momentum_buffer"].zero_()  # (a zero buffer gives the first real step what a missing one gives it)
            if hasattr(self.model, "refresh"):
                self.model.refresh()

    def _capture(self):
        self.model.train()
        # warm-up on a side stream (allocations, autotuning, momentum buffers) with the state put back afterwards
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._body()
        self._graph = g
        # the capture itself does not execute: nothing to undo

    def capture_for(self, io):
        """The same step captured on another set of input tensors (and priority output): a pipeline with several batch slots replays
        the graph of the slot at hand instead of copying the slot into the one set of static inputs (eight eager launches and
        0.15 ms of idle learner stream per step).  The graphs share one memory pool -- they never run at the same time."""
        if self._graph is None:
            self._capture()
        if getattr(self, "_pool", None) is None:
            self._pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
            self._body(io)
        return g

    def run(self):
        """One step on what the static input tensors hold (obs, action, target_reward, target_value, target_policy, weights:
        written in place by DeviceReplay.assemble / LearnerPipeline): the graph replay and nothing else -- no copy from the host,
        nothing read back.  Results stay in `self.priority` [B] and `self.out` [7] (losses, as update_weights orders them)."""
        if self._graph is None:
            self._capture()
        self._graph.replay()

    def __call__(self, batch):
        (obs_batch_ori, action_batch, mask_batch, indices, weights, make_time), (target_reward, target_value, target_policy) = batch
        cfg = self.config
        if self._graph is None:
            self._capture()
        dev = self.obs.device
        self.obs.copy_(_t(obs_batch_ori[:, 0:cfg.stacked_observations, :], dev), non_blocking=True)  # (only the first window is the model's input)
        self.action.copy_(_t(action_batch, dev, torch.long), non_blocking=True)
        self.target_reward.copy_(_t(target_reward, dev)[:, :cfg.num_unroll_steps], non_blocking=True)
        self.target_value.copy_(_t(target_value, dev), non_blocking=True)
        self.target_policy.copy_(_t(target_policy, dev), non_blocking=True)
        self.weights.copy_(_t(weights, dev), non_blocking=True)
        self._graph.replay()
        o = self.out.cpu().numpy()  # the one host round trip of a step
        loss_data = (float(o[0]), float(o[1]), float(o[2]), 0, float(o[3]), float(o[4]), float(o[5]), 0.0)
        return loss_data, self.priority.cpu().numpy()



# ---- the learner side as one device pipeline -----------------------------------------------------------------------------
class _Slot:
    """One batch in flight: the tensors DeviceReplay.assemble fills (same names as GraphedUpdate's static inputs)."""

    def __init__(self, like, R, U, A, win_shape, win_dtype):
        for k in ("obs", "action", "target_reward", "target_value", "target_policy", "weights", "priority"):
            setattr(self, k, torch.zeros_like(getattr(like, k)))
        d = like.obs.device
        self.ids = torch.zeros(like.weights.shape[0], dtype=torch.int64, device=d)
        self.value_windows = torch.zeros(win_shape, dtype=win_dtype, device=d)
        self.re_windows = torch.zeros((R * (U + 1), win_shape[1]), dtype=win_dtype, device=d) if R else None
        self.ready, self.done = torch.cuda.Event(), torch.cuda.Event()
        self.used = False
        import threading
        self.enqueued = threading.Event()   # the learner half of the step that last used this slot has been enqueued (`done` recorded)
        self.enqueued.set()


def streams_on_distinct_queues(dev, n, candidates=12, launches=150):
    """`n` PyTorch streams that run side by side.  A process has more streams than the GPU has hardware queues (4), two streams
    on one queue take turns, and which pool streams share one is not for the caller to know (measured: the two re-search streams of
    the learner pipeline, created one after the other, sat on the same queue and gained nothing over one).  So measure: a captured
    chain of small dependent GEMMs (one workgroup each, ~7 us: long against the dispatch of a launch, tiny against the GPU) on a
    candidate beside another chain on each stream chosen so far; side by side the pair takes what one takes, on one queue twice that.
    Returns (streams, ms of one chain, [ms of every pair tried])."""
    def train():
        x = torch.eye(128, device=dev)
        y = torch.eye(128, device=dev)
        for _ in range(3):
            y = x @ y
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(launches):
                y = x @ y
        return g, x, y
    torch.cuda.synchronize(dev)
    ta, tb = train(), train()   # (two graphs: launches of one and the same executable graph queue up behind each other)

    def both(a, b):
        torch.cuda.synchronize(dev)
        e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        cur = torch.cuda.current_stream(dev)
        e0.record(cur)
        for st, g, e in ((a, ta[0], ea), (b, tb[0], eb)):
            if st is None:
                continue
            st.wait_event(e0)
            with torch.cuda.stream(st):
                g.replay()
                e.record(st)
        torch.cuda.synchronize(dev)
        return max(e0.elapsed_time(ea), e0.elapsed_time(eb) if b is not None else 0.0)
    cands = [torch.cuda.Stream(device=dev) for _ in range(candidates)]
    both(cands[0], cands[1])    # (first launches of both graphs)
    both(cands[0], None)
    one = both(cands[0], None)
    chosen, pairs = [cands[0]], []
    for c in cands[1:]:
        if len(chosen) == n:
            break
        ms = [both(c, st) for st in chosen]
        pairs.append(ms)
        if max(ms) < 1.4 * one:
            chosen.append(c)
    chosen += [c for c in cands if c not in chosen][:n - len(chosen)]   # (fewer queues than asked for: the rest share)
    return chosen, one, pairs


class LearnerPipeline:
    """What the reference spreads over BatchWorker_CPU x cpu_actor, BatchWorker_GPU x gpu_actor, the BatchStorage queue and
    the learner process (/root/reference/core/train.py:317-431, 440-481; core/reanalyze_worker.py:402-440) as streams of ONE
    GPU with nothing on the host but the enqueueing:

        prepare stream   priorities of step k - 4 back into the replay -> sample ids / weights -> DeviceReplay.assemble into slot
                         k % 4 (value targets with the TARGET model) -> the re-search's inputs: everything that touches the replay
        re-search stream k % 2: policy targets for the first R rows re-searched with the TARGET model (policy_re_device) -- 768 roots
                         x 49 simulations are one latency-bound launch on a fifth of the GPU (1.4 of a batch's ~3 ms), so two
                         consecutive batches' searches run side by side
        learner stream   slot -> the captured step's static inputs -> lr -> GraphedUpdate.run -> new priorities into the slot

    so batches k + 1 .. k + 3 are in preparation while step k trains (the reference's queue of prepared batches; `slots` = batches
    in flight, by default one more than the three stages).  (research_streams=0: the re-search on the prepare stream, two slots --
    r04's first form.)
    Cadences as train.py:392-398: `on_checkpoint(step)` every checkpoint_interval steps (the caller hands the weights to the
    actors), the target model takes the learner's weights of one target_model_interval ago every target_model_interval steps.
    Nothing synchronises the host; `losses()` reads the last step's loss tuple (one small read-back) when somebody wants it.

    host_thread: the learner half is enqueued by a second host thread.  Launching the captured step costs the host 2.1 ms
    (~550 graph nodes, spent inside hipGraphLaunch without the GIL), the prepare half 2.4 ms of Python-side launches; one thread
    doing both was slower than the 3.5 ms the step takes on the GPU.  What runs on which stream, and in which order, is the same
    either way; `flush()` before touching `learn` or a slot's `done` event from outside."""

    def __init__(self, config, replay, model, target_engine, batch_size=None, reanalyze_share=0.5, amp=torch.bfloat16, beta=0.4,
                 on_checkpoint=None, seed=0, host_thread=True, research_streams=2, slots=None):
        from .device_replay import policy_re_device
        self._policy_re = policy_re_device
        self.cfg, self.replay, self.model, self.target = config, replay, model, target_engine
        self.B = int(batch_size or config.batch_size)
        self.R = int(self.B * reanalyze_share)
        self.beta, self.on_checkpoint = beta, on_checkpoint
        dev = next(model.parameters()).device
        self.optimizer = make_optimizer(model, config, capturable=True)
        self.graphed = GraphedUpdate(model, self.optimizer, config, self.B, amp=amp)
        U, A = config.num_unroll_steps, config.action_space_size
        self.Dp = target_engine.pad_observations(replay.D, replay.stack)
        win = (self.B * (U + 1), replay.stack * self.Dp)
        n_side = research_streams if self.R else 0
        picked, one_ms, pair_ms = streams_on_distinct_queues(dev, 2 + n_side)
        self.prep, self.learn, self.side = picked[0], picked[1], picked[2:]
        self.stream_probe_ms = {"one": one_ms, "pairs": pair_ms}
        n_slots = int(slots) if slots else (len(self.side) + 2 if self.side else 2)   # (one more than the stages: 465 -> 491 steps/s)
        assert n_slots >= 2
        self.slots = [_Slot(self.graphed, self.R, U, A, win, target_engine.dtype) for _ in range(n_slots)]
        self.steps = 0
        self.host_wait_s = 0.0                  # what the host spent waiting for a slot (the GPU being the slower side)
        self._recent = None
        self._re_roots = {}
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed))
        import copy
        self.net = getattr(model, "net", model)  # (the nn.Module itself where `model` is fused_train.FusedTrainNet)
        self._recent_net = copy.deepcopy(self.net)  # train.py:396-398: the target model runs one interval behind
        cur = torch.cuda.current_stream(dev)
        self.prep.wait_stream(cur)
        self.learn.wait_stream(cur)
        for st in self.side:
            st.wait_stream(cur)
        with torch.cuda.stream(self.learn):
            self.graphed._capture()             # (now, not inside the first step: a capture synchronises the device)
            for slot in self.slots:             # ... and once per batch slot, on the slot's own tensors
                slot.graph = self.graphed.capture_for(slot)
        self.prep_interference_ms = None
        if getattr(model, "_head_streams", None):
            self._pick_prepare_stream(dev)
        self._dev, self._error, self._queue, self._thread = dev, None, None, None
        if host_thread:
            import queue
            import threading
            self._queue = queue.Queue()
            self._thread = threading.Thread(target=self._learn_loop, name="hz-learner-half", daemon=True)
            self._thread.start()

    def _learn_loop(self):
        torch.cuda.set_device(self._dev)
        while True:
            item = self._queue.get()
            if item is None:
                return
            slot, k = item
            try:
                if self._error is None:
                    self._learn_half(slot, k)
            except BaseException as e:      # (surfaces in the enqueueing thread: step / flush / losses)
                self._error = e
            finally:
                slot.enqueued.set()

    def _learn_half(self, slot, k):
        with torch.cuda.stream(self.learn):
            self.learn.wait_event(slot.ready)
            adjust_lr(self.cfg, self.optimizer, k)
            slot.graph.replay()                 # (the step captured on this slot's tensors: nothing is copied in or out)
            slot.done.record(self.learn)

    def flush(self):
        """Returns once every step handed in so far has been enqueued on the streams (not: has run)."""
        for slot in self.slots:
            slot.enqueued.wait()
        if self._error is not None:
            e, self._error = self._error, None
            raise RuntimeError("the learner half of a step failed") from e

    def close(self):
        if self._thread is not None:
            self._queue.put(None)
            self._thread.join()
            self._thread = None

    def _pick_prepare_stream(self, dev, candidates=4, launches=1500):
        """A captured step with parallel branches (fused_train.FusedTrainNet's heads) replays on streams of the graph's own, and
        the GPU has fewer hardware queues (4) than the process has streams: a prepare stream that shares a queue with one of those
        branches runs its batch AFTER the step instead of beside it (measured: 156 learner steps/s against 255).  Which of
        PyTorch's pool streams collide is a matter of creation order, so measure: a captured train of small launches (about as
        long as two steps, no host in it) on each candidate beside two replays of the step; the candidate on which the pair
        finishes first becomes the prepare stream."""
        g = self.graphed
        snap = g._snapshot()
        x = torch.zeros(1024, device=dev)
        cands = [self.prep] + [torch.cuda.Stream(device=dev) for _ in range(candidates - 1)]
        train = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        with torch.cuda.graph(train, capture_error_mode="thread_local"):
            for _ in range(launches):
                x.add_(1.0)
        t = []
        for c in cands:
            torch.cuda.synchronize(dev)
            a, b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream(dev)
            a.record(cur)
            c.wait_event(a)
            self.learn.wait_event(a)
            with torch.cuda.stream(c):
                train.replay()
                b.record(c)
            with torch.cuda.stream(self.learn):
                g.run()
                g.run()
                e.record(self.learn)
            torch.cuda.synchronize(dev)
            t.append(max(a.elapsed_time(b), a.elapsed_time(e)))
        g._restore(snap)
        torch.cuda.synchronize(dev)
        self.prep = cands[min(range(len(t)), key=t.__getitem__)]
        self.prep_interference_ms = t

    def _value_fn(self, windows):
        return self.target.initial(windows, padded=self.Dp != self.replay.D)[0]

    def step(self):
        """Enqueue learner step number self.steps (both halves).  Returns nothing; never blocks the host."""
        cfg, rp, k = self.cfg, self.replay, self.steps
        ns = len(self.slots)
        slot = self.slots[k % ns]
        if self._error is not None:
            self.flush()
        if slot.used:
            import time
            slot.enqueued.wait()                # (step k - ns's learner half has been enqueued: `done` is that step's)
            t0 = time.perf_counter()
            slot.done.synchronize()             # (back-pressure: the host enqueues at most `ns` steps ahead of the one training)
            self.host_wait_s += time.perf_counter() - t0
        side = self.side[k % len(self.side)] if self.side else None
        with torch.cuda.stream(self.prep), torch.no_grad():
            if slot.used:                       # step k - ns has trained on this slot: its priorities go back, the slot is free
                self.prep.wait_event(slot.done)
                rp.update_priorities(slot.ids, slot.priority)
            if k % cfg.target_model_interval == 0 and k > 0:   # train.py:396-398
                prev = self.slots[(k - 1) % ns]
                prev.enqueued.wait()
                self.prep.wait_event(prev.done)                # (the learner's weights as of step k - 1 are complete)
                for st in self.side:
                    self.prep.wait_stream(st)                  # (no re-search with the old target model is still running)
                self.target.load(self._recent_net)
                self._recent_net.load_state_dict(self.net.state_dict())
                self._weights_taken = torch.cuda.Event()
                self._weights_taken.record(self.prep)
                self.learn.wait_event(self._weights_taken)     # (the next update must not overwrite what is being copied)
            ids, w = rp.sample(self.B, self.beta)
            slot.ids.copy_(ids)
            slot.weights.copy_(w)
            inside = rp.assemble(ids, self._value_fn, slot, value_windows=slot.value_windows, slot_elems=self.Dp)
            if self.R:
                legal, mask = rp.policy_re_inputs(ids[:self.R], slot.re_windows, slot_elems=self.Dp)
            if self.R and side is not None:
                inputs_ready = torch.cuda.Event()
                inputs_ready.record(self.prep)
            with torch.cuda.stream(side if (self.R and side is not None) else self.prep):
                if self.R:
                    where = torch.cuda.current_stream(slot.obs.device)
                    if side is not None:
                        side.wait_event(inputs_ready)
                        legal.record_stream(side)              # (allocated on the prepare stream, read here)
                        mask.record_stream(side)
                    roots = self._re_roots.get(where.cuda_stream)
                    if roots is None:                          # (one tree pool per stream that searches)
                        from . import cytree
                        roots = self._re_roots[where.cuda_stream] = cytree.Roots(slot.re_windows.shape[0], cfg.action_space_size,
                                                                                  cfg.num_simulations, device=slot.obs.device)
                    pol = self._policy_re(cfg, self.target, slot.re_windows, legal, mask, generator=self.gen, tie_seed=k,
                                          padded=self.Dp != rp.D, roots=roots)
                    slot.target_policy[:self.R] = pol.view(self.R, cfg.num_unroll_steps + 1, -1)   # [reanalyzed | stored], :412-419
                slot.ready.record(torch.cuda.current_stream(slot.obs.device))
        if self._thread is not None:
            slot.enqueued.clear()
            self._queue.put((slot, k))
        else:
            self._learn_half(slot, k)
        slot.used = True
        self.steps = k + 1
        if self.on_checkpoint is not None and self.steps % cfg.checkpoint_interval == 0:   # train.py:392-393
            slot.enqueued.wait()
            self.on_checkpoint(self.steps, slot.done)

    def losses(self):
        """(total, weighted, mean loss, 0, mean policy, mean reward, mean value, 0.0) of the last finished step (synchronises)."""
        self.flush()
        self.learn.synchronize()
        o = self.graphed.out.cpu().numpy()
        return (float(o[0]), float(o[1]), float(o[2]), 0, float(o[3]), float(o[4]), float(o[5]), 0.0)
