"""hanabizero_amd.fused_train -- the learner's forward / backward through MuZeroNet / MuZeroNetFull with every
Linear -> BatchNorm1d -> (+ skip) -> ReLU block as one hipBLASLt GEMM + ONE hand-written launch (include/hz_train.h), in 16 bits.

What it replaces: the module forward PyTorch autograd runs under autocast inside ``update_weights``
(/root/reference/core/train.py:114-222; blocks of config/hanabi_control/model.py:18-125, 131-149, 241-269).  Same parameters
(the wrapped module's own tensors: state_dict, optimiser and weight hand-over do not change), same arithmetic per block as
``F.linear`` + ``F.batch_norm(training=True)`` + add + ReLU under bf16 autocast -- 16-bit GEMM with fp32 accumulate, batch
statistics in fp32, running statistics updated with momentum 0.1, outputs rounded where autocast materialises 16-bit tensors --
but per block 2 launches forward and 4 backward instead of ~20: a learner step at batch 256 is launch-bound (~1.7 k kernels of a
few microseconds; profiles/r04_learner_kernel_stats.md).

How the gradients travel: a block's backward writes the BatchNorm affine gradients and the Linear weight gradient straight INTO
the parameters' ``.grad`` (accumulating: the recurrent blocks are used num_unroll_steps times per step), so the parameters never
enter autograd; ``.grad`` must exist and be zeroed per step (``optimizer.zero_grad(set_to_none=False)``, what GraphedUpdate
does).  The gradient of a Linear bias in front of a training-mode BatchNorm is identically zero (the normalisation removes any
per-column constant) and is left at zero rather than computed as 16-bit rounding noise.  16-bit copies of the Linear weights are
refreshed once per step (``refresh()``) instead of cast at every use.
"""
import ctypes as C

import torch
import torch.nn as nn

from ._lib import check, lib
from .model import NetworkOutput, _Dyn, _Res

_DT = {torch.bfloat16: 1, torch.float16: 2}  # include/hz_tree.h HZ_BF16 / HZ_F16


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Block:
    """Linear (+ BatchNorm1d) of the wrapped module; `w16` / `b16`: the Linear's parameters in the compute format."""

    def __init__(self, lin, bn, dtype, uses):
        self.lin, self.bn, self.uses = lin, bn, uses
        self.w16 = lin.weight.detach().to(dtype)
        self.b16 = lin.bias.detach().to(dtype)
        self.x_stack = self.dy_stack = None
        self.stack_uses = self.stack_rows = self.bwd_seen = 0
        self.scratch, self.groups = None, 1   # a call over several row groups (hz_bn_act_*_groups): what crosses the groups waits here
                                              # for hz_bn_groups_finish (FusedTrainNet._finish_groups)

    def use_stacks(self, uses, rows, k_in):
        """The block is used `uses` times per step on `rows` rows each (the dynamics net): its inputs and its pre-activation
        gradients of all uses live in two stacked buffers, so that the weight gradient is ONE GEMM over uses x rows rows when
        the last of them has been through its backward, not one per use."""
        Cn = self.lin.weight.shape[0]
        if self.x_stack is None or self.x_stack.shape != (uses * rows, k_in):
            dev, dt = self.w16.device, self.w16.dtype
            self.x_stack = torch.zeros((uses * rows, k_in), dtype=dt, device=dev)
            self.dy_stack = torch.zeros((uses * rows, Cn), dtype=dt, device=dev)
        self.stack_uses, self.stack_rows, self.bwd_seen = uses, rows, 0

    def group_scratch(self, groups):
        Cn = self.lin.weight.shape[0]
        if self.scratch is None or self.scratch.shape[0] != groups:
            self.scratch = torch.empty((groups, 2, Cn), dtype=torch.float32, device=self.w16.device)
        self.groups = groups
        return self.scratch


class _LinBNAct(torch.autograd.Function):
    """out = act(batch_norm(x @ W^T + b) + res): forward GEMM + hz_bn_act_forward; backward hz_bn_act_backward + two GEMMs."""

    @staticmethod
    def forward(ctx, x, res, blk, relu, anchor=None, groups=1, use=None, out_to=None):
        """anchor: any tensor that requires grad (the parameters do not enter autograd here: the first block of a forward, whose
        input is data, needs one for its output to be part of the graph); its gradient is None.
        groups: x stacks that many batches (the hidden states of the unrolled step's inferences): ONE GEMM over all rows, the
        BatchNorm of each batch with its own statistics -- the arithmetic of `groups` calls of the module.
        use (with _Block.use_stacks): this is the block's use number `use` of the step -- x is slot `use` of its input stack, and its
        weight gradient waits for the other uses; out_to: the block whose input stack takes this block's output (slot `use`)."""
        bn = blk.bn
        y = torch.addmm(blk.b16, x, blk.w16.t())
        rows, Cn = y.shape
        if use is not None:
            assert x.data_ptr() == blk.x_stack[use * blk.stack_rows].data_ptr() and rows == blk.stack_rows
        out = torch.empty_like(y) if out_to is None else out_to.x_stack[use * rows:(use + 1) * rows]
        assert rows % groups == 0
        B = rows // groups
        stats = torch.empty((2, groups, Cn), dtype=torch.float32, device=y.device)
        scratch = blk.group_scratch(groups) if groups > 1 else None
        if res is not None:
            assert res.shape == y.shape and res.stride(1) == 1 and res.dtype == y.dtype
        check(lib.hz_bn_act_forward_groups(y.data_ptr(), y.stride(0), None if res is None else res.data_ptr(), 0 if res is None else res.stride(0),
                                           out.data_ptr(), out.stride(0), B, groups, Cn, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                           bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps),
                                           stats[0].data_ptr(), stats[1].data_ptr(), None if scratch is None else scratch.data_ptr(),
                                           int(relu), _DT[y.dtype], _stream()),
              "hz_bn_act_forward_groups")
        ctx.blk, ctx.relu, ctx.has_res, ctx.groups, ctx.use = blk, relu, res is not None, groups, use
        ctx.save_for_backward(x, y, out, stats)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, y, out, stats = ctx.saved_tensors
        blk, bn = ctx.blk, ctx.blk.bn
        if bn.weight.grad is None or blk.lin.weight.grad is None:
            raise RuntimeError("fused_train: the blocks accumulate into existing .grad tensors -- zero them with "
                               "optimizer.zero_grad(set_to_none=False), do not drop them")
        if dout.stride(1) != 1:
            dout = dout.contiguous()
        rows, Cn = y.shape
        groups = ctx.groups
        use = ctx.use
        dy = torch.empty_like(y) if use is None else blk.dy_stack[use * rows:(use + 1) * rows]
        dres = torch.empty_like(y) if ctx.has_res else None
        scratch = blk.group_scratch(groups) if groups > 1 else None
        check(lib.hz_bn_act_backward_groups(dout.data_ptr(), dout.stride(0), out.data_ptr(), out.stride(0), y.data_ptr(), y.stride(0),
                                            dy.data_ptr(), dy.stride(0), None if dres is None else dres.data_ptr(),
                                            0 if dres is None else dres.stride(0), rows // groups, groups, Cn, bn.weight.data_ptr(),
                                            stats[0].data_ptr(), stats[1].data_ptr(), bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr(),
                                            None if scratch is None else scratch.data_ptr(), int(ctx.relu), _DT[y.dtype], _stream()),
              "hz_bn_act_backward_groups")
        g = blk.lin.weight.grad                                # W.grad += dy^T x: ONE GEMM, 16-bit operands, fp32 accumulate and output
        if use is None:
            torch.addmm(g, dy.t(), x, out_dtype=torch.float32, out=g)
        else:                                                  # ... over all uses of the step at once, when the last one is through
            blk.bwd_seen += 1
            if blk.bwd_seen == blk.stack_uses:
                torch.addmm(g, blk.dy_stack.t(), blk.x_stack, out_dtype=torch.float32, out=g)
                blk.bwd_seen = 0
        dx = torch.mm(dy, blk.w16) if ctx.needs_input_grad[0] else None
        return dx, dres, None, None, None, None, None, None


class _Lin(torch.autograd.Function):
    """A head's last layer: y = x @ W^T + b (no BatchNorm behind it)."""

    @staticmethod
    def forward(ctx, x, blk):
        ctx.blk = blk
        ctx.save_for_backward(x)
        return torch.addmm(blk.b16, x, blk.w16.t())

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        blk = ctx.blk
        g = blk.lin.weight.grad
        torch.addmm(g, dy.t(), x, out_dtype=torch.float32, out=g)
        blk.lin.bias.grad.add_(dy.sum(0, dtype=torch.float32))
        return torch.mm(dy, blk.w16), None


class _HeadLosses(torch.autograd.Function):
    """The losses of one inference (value / reward cross-entropies against the two-hot targets, policy cross-entropy against the
    visit distribution, the weighted total per row) and their gradients with respect to the three heads' logits in ONE launch
    (include/hz_train.h hz_muzero_head_losses) instead of ~20 forward and ~20 backward elementwise / softmax / reduce launches.
    Returns (row_total [B] -- differentiable --, losses [B, 4] = policy / value / reward / total, preds [B, 2] = the heads' scalars)."""

    @staticmethod
    def forward(ctx, value, reward, policy, tv, tr, tp, weights, support, coeffs):
        B, V = value.shape
        A = policy.shape[1]
        d = value.device
        dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[value.dtype]
        assert policy.dtype == value.dtype and (reward is None or reward.dtype == value.dtype)
        assert value.stride(1) == 1 and policy.stride(1) == 1 and tp.stride(1) == 1 and weights.is_contiguous()
        dv, dp = torch.empty((B, V), dtype=value.dtype, device=d), torch.empty((B, A), dtype=value.dtype, device=d)
        dr = torch.empty((B, V), dtype=value.dtype, device=d) if reward is not None else None
        losses = torch.empty((B, 4), dtype=torch.float32, device=d)
        preds = torch.empty((B, 2), dtype=torch.float32, device=d)
        vc, rc, pc = coeffs
        check(lib.hz_muzero_head_losses(value.data_ptr(), value.stride(0), None if reward is None else reward.data_ptr(),
                                        0 if reward is None else reward.stride(0), policy.data_ptr(), policy.stride(0), B, V, support.min, A, dt,
                                        tv.data_ptr(), tv.stride(0), None if tr is None else tr.data_ptr(), 0 if tr is None else tr.stride(0),
                                        tp.data_ptr(), tp.stride(0), weights.data_ptr(), float(vc), float(rc), float(pc), dv.data_ptr(),
                                        None if dr is None else dr.data_ptr(), dp.data_ptr(), losses.data_ptr(), preds.data_ptr(), _stream()),
              "hz_muzero_head_losses")
        ctx.has_reward = reward is not None
        ctx.save_for_backward(dv, dr if dr is not None else dv.new_empty(0), dp)
        ctx.mark_non_differentiable(losses, preds)
        return losses[:, 3], losses, preds

    @staticmethod
    def backward(ctx, g, _gl, _gp):
        dv, dr, dp = ctx.saved_tensors
        g = g.to(dv.dtype).unsqueeze(1)
        return dv * g, (dr * g if ctx.has_reward else None), dp * g, None, None, None, None, None, None


class _UnrolledLosses(torch.autograd.Function):
    """_HeadLosses for every inference of the unrolled step at once (include/hz_train.h hz_muzero_unrolled_losses): logits stacked
    inference by inference (value / policy [(U + 1) B, .], reward [U B, .]), targets as the learner holds them ([B, U + 1], [B, U],
    [B, U + 1, A]: indexed through their strides, nothing transposed).  Returns (row totals [(U + 1) B], losses [(U + 1) B, 4],
    preds [(U + 1) B, 2])."""

    @staticmethod
    def forward(ctx, value, reward, policy, tv, tr, tp, weights, support, coeffs):
        B = weights.shape[0]
        rows, V = value.shape
        steps = rows // B
        A = policy.shape[1]
        d = value.device
        dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[value.dtype]
        assert rows == steps * B and policy.shape[0] == rows and reward.shape[0] == rows - B and steps >= 2
        assert policy.dtype == value.dtype and reward.dtype == value.dtype
        assert value.stride(1) == 1 and reward.stride(1) == 1 and policy.stride(1) == 1 and tp.stride(2) == 1 and weights.is_contiguous()
        assert tv.shape == (B, steps) and tr.shape[0] == B and tr.shape[1] >= steps - 1 and tp.shape == (B, steps, A)
        dv, dp = torch.empty((rows, V), dtype=value.dtype, device=d), torch.empty((rows, A), dtype=value.dtype, device=d)
        dr = torch.empty((rows - B, V), dtype=value.dtype, device=d)
        losses = torch.empty((rows, 4), dtype=torch.float32, device=d)
        preds = torch.empty((rows, 2), dtype=torch.float32, device=d)
        vc, rc, pc = coeffs
        check(lib.hz_muzero_unrolled_losses(value.data_ptr(), value.stride(0), reward.data_ptr(), reward.stride(0), policy.data_ptr(),
                                            policy.stride(0), B, steps, V, support.min, A, dt, tv.data_ptr(), tv.stride(0), tv.stride(1),
                                            tr.data_ptr(), tr.stride(0), tr.stride(1), tp.data_ptr(), tp.stride(0), tp.stride(1),
                                            weights.data_ptr(), float(vc), float(rc), float(pc), dv.data_ptr(), dr.data_ptr(), dp.data_ptr(),
                                            losses.data_ptr(), preds.data_ptr(), _stream()), "hz_muzero_unrolled_losses")
        ctx.B = B
        ctx.save_for_backward(dv, dr, dp)
        ctx.mark_non_differentiable(losses, preds)
        return losses[:, 3], losses, preds

    @staticmethod
    def backward(ctx, g, _gl, _gp):
        dv, dr, dp = ctx.saved_tensors
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.float().contiguous()
        ov, orr, op = torch.empty_like(dv), torch.empty_like(dr), torch.empty_like(dp)   # d * g.to(d.dtype) for the three, one launch
        check(lib.hz_scale_rows3(dv.data_ptr(), dv.shape[1], dr.data_ptr(), dr.shape[1], dp.data_ptr(), dp.shape[1], g.data_ptr(), dv.shape[0],
                                 ctx.B, ov.data_ptr(), orr.data_ptr(), op.data_ptr(), _DT[dv.dtype], _stream()), "hz_scale_rows3")
        return ov, orr, op, None, None, None, None, None, None


class _StateAction(torch.autograd.Function):
    """[state | one-hot(action)]: the dynamics net's input (config/hanabi_control/model.py:215-219) in one launch; the gradient of the
    state is the first H columns of the output's (a view)."""

    @staticmethod
    def forward(ctx, state, action, A, use=None, out_to=None):
        B, H = state.shape
        assert state.stride(1) == 1 and action.dtype == torch.int64 and action.shape[0] == B
        out = torch.empty((B, H + A), dtype=state.dtype, device=state.device) if out_to is None else out_to.x_stack[use * B:(use + 1) * B]
        check(lib.hz_state_action_rows(state.data_ptr(), state.stride(0), action.data_ptr(), action.stride(0), B, H, A, out.data_ptr(), out.stride(0),
                                       _DT[state.dtype], _stream()), "hz_state_action_rows")
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.H], None, None, None, None


class FusedTrainNet:
    """The training-mode forward of `net` (MuZeroNet / MuZeroNetFull on a GPU) through the fused blocks.  Quacks like the module
    where the learner touches it: initial_inference / recurrent_inference (training branch of core/model.py:61-84: logits and the
    hidden state, no scalar transform), parameters / buffers / state_dict / load_state_dict / train; `net` is the module itself
    (weight hand-over: InferenceEngine.load(model.net))."""

    def __init__(self, net, dtype=torch.bfloat16, unroll_steps=5, parallel_heads=0):
        """parallel_heads (0..3): that many of the value / reward / policy heads (in this order) run on streams of their own, forked
        from and joined into the caller's (autograd runs a chain's backward on its forward's stream); inside the learner's captured
        step they become parallel branches of the hipGraph.  While every inference ran its own heads -- six short chains of ~5-us
        launches each per step -- two branches were worth 4.35 -> 3.52 ms per step (Hanabi-Full 5p, batch 256); with the heads run
        ONCE over the stacked hidden states of all inferences (compute_losses) the chains are a fifth of the step and the fork / join
        costs more than it hides: 1.74 ms without branches, 1.85 ms with two.  Hence 0; a step with branches also needs
        learner.LearnerPipeline._pick_prepare_stream (hardware queues)."""
        assert next(net.parameters()).is_cuda, "the fused blocks are HIP kernels"
        self.net, self.dtype = net, dtype
        dev0 = next(net.parameters()).device
        self.A = net.action_space_n
        U = int(unroll_steps)
        self._blocks = []
        mk = lambda lin, bn, uses: self._mk(lin, bn, uses)
        self.rep = self._chain(net._representation, 1, mk)
        d = net._dynamics_state
        self.dyn = (d.early_skip, mk(d.fc1, d.bn1, U), mk(d.fc2, d.bn2, U), mk(d.fc3, d.bn3, U))
        self.reward = self._chain(net._dynamics_reward, U, mk)
        self.actor = self._chain(net._prediction_actor, U + 1, mk)
        self.value = self._chain(net._prediction_value, U + 1, mk)
        n_side = max(0, min(3, int(parallel_heads or 0)))
        self._head_streams = [torch.cuda.Stream(device=dev0) for _ in range(n_side)] + [None] * (3 - n_side) if n_side else None
        for p in net.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        self._w32 = [b.lin.weight for b in self._blocks] + [b.lin.bias for b in self._blocks]
        self._w16 = [b.w16 for b in self._blocks] + [b.b16 for b in self._blocks]
        self._anchor = torch.zeros(1, device=self._w16[0].device, requires_grad=True)
        self._finish_tables = {}
        self._counters = [b.bn.num_batches_tracked for b in self._blocks if b.bn is not None]
        self._uses = [b.uses for b in self._blocks if b.bn is not None]

    def _mk(self, lin, bn, uses):
        b = _Block(lin, bn, self.dtype, uses)
        self._blocks.append(b)
        return b

    def _chain(self, seq, uses, mk):
        """nn.Sequential of [Linear, BatchNorm1d, ReLU]* / _Res / trailing Linear -> a list of steps."""
        mods, steps, i = list(seq), [], 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear) and i + 2 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d):
                assert isinstance(mods[i + 2], nn.ReLU)
                steps.append(("lbr", mk(m, mods[i + 1], uses)))
                i += 3
            elif isinstance(m, _Res):
                steps.append(("res", m.early_skip, mk(m.fc1, m.bn1, uses), mk(m.fc2, m.bn2, uses)))
                i += 1
            elif isinstance(m, nn.Linear):
                steps.append(("lin", mk(m, None, uses)))
                i += 1
            else:
                raise TypeError("fused_train: unexpected module %r" % (m,))
        return steps

    # -- the pieces the learner touches -------------------------------------------------------------------------------------
    def parameters(self):
        return self.net.parameters()

    def buffers(self):
        return self.net.buffers()

    def state_dict(self):
        return self.net.state_dict()

    def load_state_dict(self, sd):
        out = self.net.load_state_dict(sd)
        self.refresh()
        return out

    def train(self, mode=True):
        self.net.train(mode)
        return self

    def refresh(self):
        """The 16-bit copies of the Linear parameters from the fp32 ones (once per optimiser step; one multi-tensor launch)."""
        with torch.no_grad():
            torch._foreach_copy_(self._w16, self._w32)

    fused_heads = True  # learner.compute_losses hands the whole unrolled forward + losses to compute_losses() below

    def compute_losses(self, config, obs_batch, action_batch, target_reward, target_value, target_policy, weights):
        """learner.compute_losses (core/train.py:114-222).  Same returns.  The three heads run ONCE over the stacked hidden states
        of all 1 + U inferences (they depend on nothing but their own state), every inference's losses, priorities' ingredients and
        logit gradients come from one launch (_UnrolledLosses)."""
        U = config.num_unroll_steps
        B = obs_batch.shape[0]
        vs, rs = config.value_support, config.reward_support
        assert (vs.min, vs.size) == (rs.min, rs.size)
        coeffs = (config.value_loss_coeff, config.reward_loss_coeff, config.policy_loss_coeff)
        # the hidden states of all inferences first (representation, then the dynamics net step by step), ...
        state = self._run(self.rep, obs_batch.reshape(B, -1).to(self.dtype))
        states = [state]
        for k in range(U):
            state = self._dynamics(state, action_batch[:, k:k + 1], k, U)
            state.register_hook(lambda grad: grad * 0.5)  # train.py:169 (the hook sees the heads' share and the next step's)
            states.append(state)
        stacked = torch.cat(states, 0)                    # [(U + 1) B, H]
        # ... then every head ONCE over the stack -- one GEMM per layer instead of one per layer and inference, each inference's
        # BatchNorm with its own statistics (hz_bn_act_*_groups) -- and all inferences' losses in one launch
        value, reward, policy_logits = self._heads(stacked, True, groups=U + 1, reward_from=B)
        # what crosses the groups of the heads' BatchNorms -- running statistics now, the affine gradients once the heads' backward
        # has run (the gradient of `stacked` is complete exactly then) -- in one launch each (hz_bn_groups_finish)
        self._finish_groups(backward=False)
        if stacked.requires_grad:
            stacked.register_hook(self._finish_groups_backward)
        tot, L, P = _UnrolledLosses.apply(value, reward, policy_logits, target_value, target_reward, target_policy, weights, vs, coeffs)
        weighted_loss = tot.sum()
        Ls = L.view(U + 1, B, 4).sum(0)
        value_priority = (P[:B, 0] - target_value[:, 0]).abs()
        reward_priority = (P[B:, 1].view(U, B) - target_reward[:, :U].t()).abs().mean(0)
        vc, rc, pc = coeffs
        return weighted_loss, dict(loss=pc * Ls[:, 0] + vc * Ls[:, 1] + rc * Ls[:, 2], policy_loss=Ls[:, 0], value_loss=Ls[:, 1],
                                   reward_loss=Ls[:, 2], value_priority=value_priority, reward_priority=reward_priority)

    def compute_losses_stepwise(self, config, obs_batch, action_batch, target_reward, target_value, target_policy, weights):
        """compute_losses inference by inference, as the module itself is called (initial_inference, then recurrent_inference per
        unroll step; one _HeadLosses launch each): the form the stacked one is tested against -- same arithmetic per batch row,
        up to the GEMMs' own summation order at another row count."""
        U = config.num_unroll_steps
        B = obs_batch.shape[0]
        vs, rs = config.value_support, config.reward_support
        coeffs = (config.value_loss_coeff, config.reward_loss_coeff, config.policy_loss_coeff)
        value, _, policy_logits, hidden_state = self.initial_inference(obs_batch.reshape(B, -1))
        tot, L, P = _HeadLosses.apply(value, None, policy_logits, target_value[:, 0], None, target_policy[:, 0], weights, vs, coeffs)
        value_priority = (P[:, 0] - target_value[:, 0]).abs()
        totals, Ls, reward_priority = [tot], [L], []
        for k in range(U):
            value, reward, policy_logits, hidden_state = self.recurrent_inference(hidden_state, action_batch[:, k:k + 1])
            tot, L, P = _HeadLosses.apply(value, reward, policy_logits, target_value[:, k + 1], target_reward[:, k], target_policy[:, k + 1],
                                          weights, vs, coeffs)
            hidden_state.register_hook(lambda grad: grad * 0.5)  # train.py:169
            totals.append(tot)
            Ls.append(L)
            reward_priority.append((P[:, 1] - target_reward[:, k]).abs())
        weighted_loss = torch.stack(totals).sum()
        Ls = torch.stack(Ls).sum(0)
        vc, rc, pc = coeffs
        return weighted_loss, dict(loss=pc * Ls[:, 0] + vc * Ls[:, 1] + rc * Ls[:, 2], policy_loss=Ls[:, 0], value_loss=Ls[:, 1],
                                   reward_loss=Ls[:, 2], value_priority=value_priority, reward_priority=torch.stack(reward_priority).mean(0))

    def _finish_groups(self, backward):
        from ._lib import BnFinish
        blocks = [b for b in self._blocks if b.bn is not None and b.groups > 1 and b.scratch is not None]
        if not blocks:
            return
        if backward:
            ptrs = [(b.bn.bias.grad.data_ptr(), b.bn.weight.grad.data_ptr(), b.scratch.data_ptr(), b.scratch.shape[2], b.groups, 0.0) for b in blocks]
        else:
            ptrs = [(b.bn.running_mean.data_ptr(), b.bn.running_var.data_ptr(), b.scratch.data_ptr(), b.scratch.shape[2], b.groups, float(b.bn.momentum))
                    for b in blocks]
        cached = self._finish_tables.get(backward)
        if cached is None or cached[0] != ptrs:   # (a table of pointers: rebuilt only if a buffer has moved -- never inside a captured step)
            arr = (BnFinish * len(ptrs))(*[BnFinish(dst0=a, dst1=b, scratch=s, cols=c, groups=g, momentum=m) for a, b, s, c, g, m in ptrs])
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(blocks[0].scratch.device)
            cached = self._finish_tables[backward] = (ptrs, dev, max(p[3] for p in ptrs))
        check(lib.hz_bn_groups_finish(cached[1].data_ptr(), len(ptrs), cached[2], int(backward), _stream()), "hz_bn_groups_finish")

    def _finish_groups_backward(self, grad):
        self._finish_groups(backward=True)
        return None

    def count_batches(self):
        """num_batches_tracked of every BatchNorm as the module's own forward would have left it after one learner step."""
        with torch.no_grad():
            torch._foreach_add_(self._counters, self._uses)

    # -- forward ------------------------------------------------------------------------------------------------------------
    def _run(self, steps, x, groups=1):
        for st in steps:
            if st[0] == "lbr":
                x = _LinBNAct.apply(x, None, st[1], True, None if x.requires_grad else self._anchor, groups)
            elif st[0] == "res":
                _, early, b1, b2 = st
                if early:   # ResMLP: skip added behind the first BatchNorm (model.py:18-30)
                    x = _LinBNAct.apply(_LinBNAct.apply(x, x, b1, True, None, groups), None, b2, True, None, groups)
                else:       # NewResMLP: behind the second (model.py:43-57)
                    x = _LinBNAct.apply(_LinBNAct.apply(x, None, b1, True, None, groups), x, b2, True, None, groups)
            else:
                x = _Lin.apply(x, st[1])
        return x

    def _heads(self, state, with_reward, groups=1, reward_from=0):
        """(value logits, reward logits or None, policy logits) of a hidden state -- or of `groups` stacked batches of hidden
        states, the reward head from row `reward_from` on (the initial inference has none)."""
        chains = [self.value, self.reward if with_reward else None, self.actor]
        rows = state.shape[0] // groups
        inputs = [(state, groups), (state[reward_from:], groups - reward_from // rows), (state, groups)]
        if self._head_streams is None:
            return [None if c is None else self._run(c, *i) for c, i in zip(chains, inputs)]
        cur = torch.cuda.current_stream(state.device)
        outs = []
        for c, st, i in zip(chains, self._head_streams, inputs):
            if c is None or st is None:
                outs.append(None if c is None else self._run(c, *i))
                continue
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(self._run(c, *i))
        for c, st in zip(chains, self._head_streams):
            if c is not None and st is not None:
                cur.wait_stream(st)
        return outs

    def initial_inference(self, obs):  # core/model.py:61-71, training branch
        state = self._run(self.rep, obs.to(self.dtype))
        value, _, policy = self._heads(state, False)
        return NetworkOutput(value, [0.0] * obs.shape[0], policy, state)

    def _dynamics(self, hidden_state, action, use=None, uses=0):
        """config/hanabi_control/model.py:61-125 behind the one-hot concat of :215-219.  use / uses: step `use` of the `uses` unroll
        steps of a learner step -- the three blocks keep their inputs and pre-activation gradients in stacks and take their weight
        gradients in one GEMM each over all steps (_Block.use_stacks)."""
        early, b1, b2, b3 = self.dyn
        if use is None:
            sa = _StateAction.apply(hidden_state, action, self.A)
            y = _LinBNAct.apply(sa, hidden_state if early else None, b1, True)
            y = _LinBNAct.apply(y, None, b2, True)
            return _LinBNAct.apply(y, None if early else hidden_state, b3, True)
        if use == 0:
            B, H = hidden_state.shape
            b1.use_stacks(uses, B, H + self.A)
            b2.use_stacks(uses, B, b1.lin.weight.shape[0])
            b3.use_stacks(uses, B, b2.lin.weight.shape[0])
        sa = _StateAction.apply(hidden_state, action, self.A, use, b1)
        y = _LinBNAct.apply(sa, hidden_state if early else None, b1, True, None, 1, use, b2)
        y = _LinBNAct.apply(y, None, b2, True, None, 1, use, b3)
        return _LinBNAct.apply(y, None if early else hidden_state, b3, True, None, 1, use, None)

    def recurrent_inference(self, hidden_state, action):  # core/model.py:74-84, training branch
        state = self._dynamics(hidden_state, action)
        value, reward, policy = self._heads(state, True)
        return NetworkOutput(value, reward, policy, state)
