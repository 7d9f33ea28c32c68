"""hanabizero_amd.model -- the representation / dynamics / prediction networks of HanabiZero in PyTorch-ROCm.

Architecture and parameter names follow /root/reference/config/hanabi_control/model.py (``MuZeroNet`` :127-156 for
Hanabi-Small, ``MuZeroNetFull`` :237-276 for Hanabi-Full) and core/model.py:46-97 (``BaseMuZeroNet``), so a reference
``state_dict`` loads unchanged (``set_weights`` / ``get_weights``) and core/train.py can keep training this module.

What is different is the inference side, which the self-play hot path calls once per tree depth over the whole
envs batch: ``InferenceEngine`` folds every eval-mode BatchNorm1d into its Linear, keeps weights pre-cast
(bf16 / fp16 / fp32) and pre-transposed in HBM, replaces the one-hot concat of ``dynamics`` (model.py:215-219) by a
column lookup of the first layer's action block, runs the three heads' first layers as one GEMM, and returns DEVICE
tensors (the reference returns numpy on the host: core/model.py:65-71,78-82).  The GEMMs go to hipBLASLt/rocBLAS
(MFMA); nothing else on the hot path touches the matrix cores.
"""
import math
import os
from typing import NamedTuple

import torch
import torch.nn as nn
import torch.nn.functional as F


class NetworkOutput(NamedTuple):  # core/model.py:13-17
    value: object
    reward: object
    policy_logits: object
    hidden_state: object


# ------------------------------------------------------------------------------------------------ transforms
def inverse_scalar_transform(logits, support_min, support_max, epsilon=0.001):
    """core/config.py:210-232 (delta = 1): categorical logits over integers [min, max] -> scalar via h^-1."""
    probs = torch.softmax(logits.float(), dim=1)
    support = torch.arange(support_min, support_max + 1, dtype=probs.dtype, device=probs.device)
    value = (probs * support).sum(1, keepdim=True)
    sign = torch.where(value < 0, -torch.ones_like(value), torch.ones_like(value))
    out = ((torch.sqrt(1 + 4 * epsilon * (torch.abs(value) + 1 + epsilon)) - 1) / (2 * epsilon)) ** 2 - 1
    out = sign * out
    return torch.nan_to_num(out, nan=0.0, posinf=float("inf"), neginf=float("-inf"))


# ------------------------------------------------------------------------------------------------ blocks
def _lbr(i, o):
    return [nn.Linear(i, o), nn.BatchNorm1d(o), nn.ReLU()]


class _Res(nn.Module):
    """Two Linear+BatchNorm layers with a skip connection.  early_skip=True is the reference's ``ResMLP``
    (skip added after the first BN, model.py:18-30); False is ``NewResMLP`` (after the second, model.py:43-57)."""

    def __init__(self, dim, early_skip):
        super().__init__()
        self.in_dim, self.early_skip = dim, early_skip
        self.fc1, self.bn1 = nn.Linear(dim, dim), nn.BatchNorm1d(dim)
        self.fc2, self.bn2 = nn.Linear(dim, dim), nn.BatchNorm1d(dim)

    def forward(self, x):
        y = self.bn1(self.fc1(x))
        if self.early_skip:
            y = y + x
        y = self.bn2(self.fc2(F.relu(y)))
        if not self.early_skip:
            y = y + x
        return F.relu(y)


class _Dyn(nn.Module):
    """Three Linear+BatchNorm layers over [state | one-hot action]; the state is added back after the first
    (``DynamicNet`` model.py:61-91) or the last (``NewDynamicNet`` model.py:93-125) of them."""

    def __init__(self, state_dim, action_dim, early_skip):
        super().__init__()
        self.state_dim, self.action_dim, self.early_skip = state_dim, action_dim, early_skip
        self.fc1, self.bn1 = nn.Linear(state_dim + action_dim, state_dim), nn.BatchNorm1d(state_dim)
        self.fc2, self.bn2 = nn.Linear(state_dim, state_dim), nn.BatchNorm1d(state_dim)
        self.fc3, self.bn3 = nn.Linear(state_dim, state_dim), nn.BatchNorm1d(state_dim)

    def forward(self, state_action):
        state = state_action[:, :self.state_dim]
        y = self.bn1(self.fc1(state_action))
        if self.early_skip:
            y = y + state
        y = F.relu(y)
        y = F.relu(self.bn2(self.fc2(y)))
        y = self.bn3(self.fc3(y))
        if not self.early_skip:
            y = y + state
        return F.relu(y)


class _HanabiNet(nn.Module):
    """BaseMuZeroNet (core/model.py:46-103) for both games; subclasses only lay out the sub-networks."""

    feature_size = 512

    def __init__(self, action_space_n, inverse_value_transform, inverse_reward_transform, state_norm=False):
        super().__init__()
        assert not state_norm, "state_norm is off in both Hanabi configs (config/hanabi_control/__init__.py:37,155)"
        self.action_space_n = action_space_n
        self.inverse_value_transform = inverse_value_transform
        self.inverse_reward_transform = inverse_reward_transform
        self.state_norm = state_norm

    def _zero_heads(self):  # model.py:151-156 / :271-276
        for head in (self._prediction_value, self._dynamics_reward, self._prediction_actor):
            nn.init.zeros_(head[-1].weight)
            nn.init.zeros_(head[-1].bias)

    def representation(self, obs_history):
        return self._representation(obs_history)

    def prediction(self, state):
        return self._prediction_actor(state), self._prediction_value(state)

    def dynamics(self, state, action):
        assert state.dim() == 2 and action.shape[1] == 1
        one_hot = torch.zeros(action.shape[0], self.action_space_n, dtype=torch.float32, device=action.device)
        one_hot.scatter_(1, action, 1.0)
        next_state = self._dynamics_state(torch.cat((state, one_hot), dim=1))
        return next_state, self._dynamics_reward(next_state)

    def initial_inference(self, obs):  # core/model.py:61-71
        state = self.representation(obs)
        logits, value = self.prediction(state)
        if not self.training:
            value = self.inverse_value_transform(value).detach().cpu().numpy()
            state = state.detach().cpu().numpy()
            logits = logits.detach().cpu().numpy()
        return NetworkOutput(value, [0. for _ in range(obs.size(0))], logits, state)

    def recurrent_inference(self, hidden_state, action):  # core/model.py:74-84
        state, reward = self.dynamics(hidden_state, action)
        logits, value = self.prediction(state)
        if not self.training:
            value = self.inverse_value_transform(value).detach().cpu().numpy()
            reward = self.inverse_reward_transform(reward).detach().cpu().numpy()
            state = state.detach().cpu().numpy()
            logits = logits.detach().cpu().numpy()
        return NetworkOutput(value, reward, logits, state)

    def get_weights(self):
        return {k: v.cpu() for k, v in self.state_dict().items()}

    def set_weights(self, weights):
        self.load_state_dict(weights)

    def get_params_mean(self):
        return 0, 0, 0, 0


class MuZeroNet(_HanabiNet):
    """Hanabi-Small network (model.py:127-156): feature 512, head hidden 128."""

    def __init__(self, input_size, action_space_n, reward_support_size, value_support_size, inverse_value_transform,
                 inverse_reward_transform, state_norm=False, proj=False):
        super().__init__(action_space_n, inverse_value_transform, inverse_reward_transform, state_norm)
        assert not proj, "the projection heads are asserted off in the reference (model.py:160-161)"
        f, h = self.feature_size, 128
        self.hidden_size = h
        self._representation = nn.Sequential(*_lbr(input_size, f), _Res(f, early_skip=True))
        self._dynamics_state = _Dyn(f, action_space_n, early_skip=True)
        self._dynamics_reward = nn.Sequential(*_lbr(f, h), nn.Linear(h, reward_support_size))
        self._prediction_actor = nn.Sequential(*_lbr(f, h), nn.Linear(h, action_space_n))
        self._prediction_value = nn.Sequential(*_lbr(f, h), nn.Linear(h, value_support_size))
        self._zero_heads()


class MuZeroNetFull(_HanabiNet):
    """Hanabi-Full network (model.py:237-276): representation 1024 -> 512, head hidden 256."""

    def __init__(self, input_size, action_space_n, reward_support_size, value_support_size, inverse_value_transform,
                 inverse_reward_transform, state_norm=False):
        super().__init__(action_space_n, inverse_value_transform, inverse_reward_transform, state_norm)
        f, i, h = self.feature_size, 1024, 256
        self.init_size, self.hidden_size = i, h
        self._representation = nn.Sequential(*_lbr(input_size, i), _Res(i, early_skip=False), *_lbr(i, f),
                                             _Res(f, early_skip=False))
        self._dynamics_state = _Dyn(f, action_space_n, early_skip=False)
        self._dynamics_reward = nn.Sequential(*_lbr(f, h), *_lbr(h, h), nn.Linear(h, reward_support_size))
        self._prediction_actor = nn.Sequential(*_lbr(f, h), _Res(h, early_skip=False), nn.Linear(h, action_space_n))
        self._prediction_value = nn.Sequential(*_lbr(f, h), *_lbr(h, h), nn.Linear(h, value_support_size))
        self._zero_heads()


# ------------------------------------------------------------------------------------------------ inference engine
def _fold(linear, bn=None):
    """eval-mode BatchNorm1d(Linear(x)) == x @ W'^T + b' (fp32 fold)."""
    w, b = linear.weight.detach().float(), linear.bias.detach().float()
    if bn is None:
        return w, b
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    return w * s[:, None], (b - bn.running_mean.detach().float()) * s + bn.bias.detach().float()


_FUSED_ACT = hasattr(torch, "_addmm_activation")


class _Lin:
    """y = x @ W^T + b with W [out, in] resident in the engine's dtype and handed to the GEMM as its transposed view (the
    "TN" form hipBLASLt runs 15-20 % faster than a pre-transposed [in, out] copy for these shapes); optional in-place
    ReLU.  `put` = InferenceEngine._put: the device tensors are allocated by the first load() and overwritten in place by
    later ones."""

    def __init__(self, w, b, put):
        self.w = put(w)
        self.wt = self.w.t()
        self.b = put(b)

    def __call__(self, x, relu=False, out=None):
        if relu and out is None and _FUSED_ACT:
            return torch._addmm_activation(self.b, x, self.wt, use_gelu=False)  # bias + ReLU in the GEMM epilogue
        y = torch.addmm(self.b, x, self.wt, out=out) if out is not None else torch.addmm(self.b, x, self.wt)
        return torch.relu_(y) if relu else y


class InferenceEngine:
    """Device-resident eval-mode forward of a MuZeroNet / MuZeroNetFull.

    initial(obs [N, stack*D])                      -> value [N] f32, logits [N, A] f32, hidden [N, 512] (engine dtype)
    recurrent(hidden [N, 512], action [N] int)     -> value [N] f32, reward [N] f32, logits [N, A] f32, hidden
    `hidden_out=` lets the caller point the new hidden state at its slot of the search's hidden-state pool.
    """

    def __init__(self, net, support, dtype=torch.bfloat16, device="cuda", fused=None):
        """fused: run the search loop's recurrent inference as the single hand-written MFMA kernel of
        include/hz_mlp.h (default: whenever the engine is bf16 on a GPU); False keeps the hipBLASLt GEMM chain.
        fused="fp16x2" (dtype float32): the engine inside the contract's 1e-3 of the reference's fp32 nets with a hand-written
        kernel behind its search -- states, pool and the root inference's GEMMs stay fp32, the recurrent inference is the
        MFMA kernel in its fp16-pair build (HZ_F16X2: every fp32 number as hi + lo halves, three MFMAs per product)."""
        self.split = fused == "fp16x2"
        assert not self.split or dtype == torch.float32, "fused='fp16x2' is the hand-written recurrent inference of an fp32 engine"
        fused = True if self.split else fused
        self.dtype, self.device = dtype, torch.device(device)
        self.A, self.H = net.action_space_n, net.feature_size
        self.support = int(support)
        self.full = isinstance(net, MuZeroNetFull)
        self.use_fused = (dtype in (torch.bfloat16, torch.float16) and self.device.type == "cuda") if fused is None else bool(fused)
        self.fused = None
        self._obs_pad = None  # (D, Dp, stack) once pad_observations() has been asked for
        self._dev, self._seq = {}, 0  # device tensors by creation order within a load()
        self._fused_shapes, self.fused_tail = {}, None
        # (the stand-alone kernel's shape; the fp16-pair build streams twice the weights and hides their latency better with eight
        # wavefronts: 77 us against 84 us per 4096-row inference of the Hanabi-Full nets)
        self._fused_default = (8, 4) if self.split else (4, 4)
        self.version = 0
        self.load(net)

    def _put(self, t, dtype=None, name=None):
        """The device-resident copy of host tensor `t`: allocated by the first load(), overwritten IN PLACE by every later
        one.  A hipGraph captured over this engine (SelfPlayActor._capture) has the tensors' addresses baked in, so a
        weight update must never move them: engine.load(net) after net.set_weights(...) is all a running actor needs."""
        if name is None:
            name, self._seq = "t%d" % self._seq, self._seq + 1
        t = t.detach().to(dtype or self.dtype).contiguous()
        cur = self._dev.get(name)
        if cur is None:
            cur = self._dev[name] = t.to(self.device)
        else:
            assert cur.shape == t.shape and cur.dtype == t.dtype, "load(): %s changed shape %s -> %s" % (name, tuple(cur.shape), tuple(t.shape))
            cur.copy_(t)
        return cur

    def load(self, net):
        """(Re)build the folded weights from `net` (call after set_weights: selfplay_worker.py:177-184).  Every device
        tensor keeps its address (see _put), so captured graphs and concurrent readers on this stream see the new weights
        from their next replay on."""
        self._seq = 0
        L = lambda lin, bn=None: _Lin(*_fold(lin, bn), self._put)
        rep, dyn = net._representation, net._dynamics_state
        rw, ac, va = net._dynamics_reward, net._prediction_actor, net._prediction_value
        if self.full:
            self.rep = [L(rep[0], rep[1]), L(rep[3].fc1, rep[3].bn1), L(rep[3].fc2, rep[3].bn2), L(rep[4], rep[5]),
                        L(rep[7].fc1, rep[7].bn1), L(rep[7].fc2, rep[7].bn2)]
        else:
            self.rep = [L(rep[0], rep[1]), L(rep[3].fc1, rep[3].bn1), L(rep[3].fc2, rep[3].bn2)]
        self._rep0_folded = _fold(rep[0], rep[1])
        self.rep0p = None
        if self._obs_pad is not None:
            self.pad_observations(self._obs_pad[0], self._obs_pad[2])
        # dynamics layer 1: split [W_state | W_action]; the one-hot product is a row lookup of W_action^T
        w1, b1 = _fold(dyn.fc1, dyn.bn1)
        self.dyn1 = _Lin(w1[:, :self.H], b1, self._put)
        self.dyn1_act = self._put(w1[:, self.H:].t())  # [A, H]
        # ... and the same layer over the [state | one-hot | 0-pad] rows the traverse kernel writes (K padded to 32)
        self.onehot_cols = ((self.A + 31) // 32) * 32
        w1p = torch.zeros(w1.shape[0], self.H + self.onehot_cols, device=w1.device)
        w1p[:, :w1.shape[1]] = w1
        self.dyn1p = _Lin(w1p, b1, self._put)
        self.dyn2, self.dyn3 = L(dyn.fc2, dyn.bn2), L(dyn.fc3, dyn.bn3)
        # the three heads' first layers share their input: one GEMM [512 -> 3h] (reward | actor | value)
        ws, bs = zip(_fold(rw[0], rw[1]), _fold(ac[0], ac[1]), _fold(va[0], va[1]))
        self.h = ws[0].shape[0]
        self.heads1 = _Lin(torch.cat(ws, 0), torch.cat(bs, 0), self._put)
        self.pred1 = _Lin(torch.cat(ws[1:], 0), torch.cat(bs[1:], 0), self._put)
        # search-loop form: each head block is h real columns + 32 pad columns whose first one is a constant 1
        # (zero weights, bias 1, ReLU(1) = 1): it carries the next layers' biases through plain batched GEMMs
        self.hp = self.h + 32
        wpad, bpad = [], []
        for w, b in zip(ws, bs):
            wpad.append(torch.cat((w, torch.zeros(32, w.shape[1], device=w.device)), 0))
            one = torch.zeros(32, device=w.device)
            one[0] = 1.0
            bpad.append(torch.cat((b, one), 0))
        self.heads1p = _Lin(torch.cat(wpad, 0), torch.cat(bpad, 0), self._put)
        if self.full:
            self.rw_tail = [L(rw[3], rw[4]), L(rw[6])]
            self.ac_tail = [L(ac[3].fc1, ac[3].bn1), L(ac[3].fc2, ac[3].bn2), L(ac[4])]
            self.va_tail = [L(va[3], va[4]), L(va[6])]
            # batched tails (reward | actor | value): layer 2 = three [h -> h] GEMMs, layer 3 = three [h -> h]
            # GEMMs with the 2*support+1 wide outputs zero-padded to h
            self.bw2 = self._stack([_fold(rw[3], rw[4]), _fold(ac[3].fc1, ac[3].bn1), _fold(va[3], va[4])], self.hp, True)
            self.bw3 = self._stack([_fold(rw[6]), _fold(ac[3].fc2, ac[3].bn2), _fold(va[6])], self.h, False)
        else:
            self.rw_tail, self.ac_tail, self.va_tail = [L(rw[3])], [L(ac[3])], [L(va[3])]
            self.out_pad = ((max(2 * self.support + 1, self.A) + 31) // 32) * 32
            self.bw3 = self._stack([_fold(rw[3]), _fold(ac[3]), _fold(va[3])], self.out_pad, False)
        self.V = 2 * self.support + 1
        self._net = net
        if self.use_fused:
            if not self._fused_shapes:
                self._fused_shapes[self._fused_default] = FusedRecurrent(net, self, *self._fused_default)
                # (the root inference's tail in the 16 x 2 shape, the one that has the arrival counters; fp32 engine: GEMMs)
                self.fused_tail = FusedInitialTail(net, self, 16, 2) if self.full and not self.split else None
            else:  # same job tables, new numbers: into the tensors the kernels (and captured graphs) already point at
                for (waves, tiles), chain in self._fused_shapes.items():
                    chain.reload(FusedRecurrent(net, self, waves, tiles, values_only=True))
                if self.fused_tail is not None:
                    self.fused_tail.reload(FusedInitialTail(net, self, 16, 2, values_only=True))
            self.fused = self._fused_shapes[self._fused_default]
        self.version += 1

    def pad_observations(self, D, stack, multiple=8):
        """Lay the first representation layer out for observation windows whose `stack` slots are padded from D to
        Dp = ceil(D / multiple) * multiple elements (zero weights on the pad columns): rows of stack * Dp elements start
        on 16-byte boundaries in bf16, which is what lets hipBLASLt pick its vectorised kernels (K = 3132 -> 3136 for
        Hanabi-Full: 44 -> 30 us for that GEMM) and the actor move whole 16-byte pieces when it shifts the window.
        Returns Dp; `initial(obs, padded=True)` then takes [N, stack * Dp] rows."""
        Dp = (D + multiple - 1) // multiple * multiple
        w, b = self._rep0_folded
        assert w.shape[1] == stack * D, (w.shape, stack, D)
        wp = torch.zeros(w.shape[0], stack, Dp, device=w.device)
        wp[:, :, :D] = w.view(w.shape[0], stack, D)
        put = lambda t, _names=iter(("rep0p.w", "rep0p.b")): self._put(t, name=next(_names))
        self.rep0p = _Lin(wp.reshape(w.shape[0], stack * Dp), b, put)
        self._obs_pad = (D, Dp, stack)
        return Dp

    def fused_shape(self, waves, tiles):
        """The fused recurrent inference laid out for another workgroup shape (16 x 2: the persistent search kernel);
        built on first use, refreshed in place by every later load()."""
        if not self.use_fused:
            return None
        if (waves, tiles) not in self._fused_shapes:
            self._fused_shapes[(waves, tiles)] = FusedRecurrent(self._net, self, waves, tiles)
        return self._fused_shapes[(waves, tiles)]

    def _stack(self, folded, out_width, carry_one):
        """[(W [o, h], b [o])] * 3 -> Wt [3, hp, out_width] in the engine dtype for torch.bmm over the three head
        blocks: rows 0..h-1 = W^T, row h = b (multiplied by the constant-one input column), other rows 0; with
        carry_one the output keeps a constant-one column at index h for the next layer's bias."""
        out = torch.zeros(3, self.hp, out_width, device=folded[0][0].device)
        for k, (w, b) in enumerate(folded):
            out[k, :w.shape[1], :w.shape[0]] = w.t()
            out[k, self.h, :b.shape[0]] = b
            if carry_one:
                out[k, self.h, self.h] = 1.0
        return self._put(out)

    # -- pieces ---------------------------------------------------------------------------------------
    def _scalar(self, logits):
        if logits.is_cuda and logits.stride(1) == 1:  # one HIP kernel instead of ~25 elementwise launches
            return self.support_to_scalar(logits)
        return inverse_scalar_transform(logits, -self.support, self.support).reshape(-1)

    def _representation(self, x, first=None):
        r = self.rep
        first = first or r[0]
        if self.full:  # Linear-BN-ReLU, NewResMLP(1024), Linear-BN-ReLU, NewResMLP(512)
            x = first(x, relu=True)
            x = self._add_relu(r[2](r[1](x, relu=True)), x)
            x = r[3](x, relu=True)
            return self._add_relu(r[5](r[4](x, relu=True)), x)
        x = first(x, relu=True)  # Linear-BN-ReLU, ResMLP(512)
        y = torch.relu_(r[1](x) + x)
        return r[2](y, relu=True)

    def _dynamics(self, state, action, out=None):
        y = self.dyn1(state) + self.dyn1_act.index_select(0, action)
        if self.full:
            y = self.dyn2(torch.relu_(y), relu=True)
            y = self.dyn3(y, out=out)
            return torch.relu_(y.add_(state))
        y = torch.relu_(y.add_(state))
        y = self.dyn2(y, relu=True)
        return self.dyn3(y, relu=True, out=out)

    def _tails(self, z, with_reward):
        h = self.h
        off = h if with_reward else 0
        za, zv = z[:, off:off + h], z[:, off + h:off + 2 * h]
        if self.full:
            a = self.ac_tail
            ya = self._add_relu(a[1](a[0](za, relu=True)), za)
            logits = a[2](ya)
            value = self.va_tail[1](self.va_tail[0](zv, relu=True))
            reward = self.rw_tail[1](self.rw_tail[0](z[:, :h], relu=True)) if with_reward else None
        else:
            logits, value = self.ac_tail[0](za), self.va_tail[0](zv)
            reward = self.rw_tail[0](z[:, :h]) if with_reward else None
        return logits.float(), value, reward

    # -- entry points -----------------------------------------------------------------------------------
    @torch.no_grad()
    def initial(self, obs, state_out=None, padded=False):
        """(value [N] f32, policy logits [N, A] f32, hidden state [N, H]).  state_out: optional [N, H] buffer of the
        engine dtype for the hidden state (e.g. plane 0 of the search's pool: saves the copy there).  padded: obs rows are
        in the slot-padded layout of pad_observations()."""
        first = self.rep0p if padded else self.rep[0]
        assert first is not None, "call pad_observations() first"
        if self.fused_tail is not None and obs.is_cuda:
            # the two large representation layers as GEMMs, everything after them in one launch of the MFMA kernel
            r, x = self.rep, obs.to(self.dtype)
            x = first(x, relu=True)
            y = r[2](r[1](x, relu=True))   # (the block's skip -- relu(y + x) -- is added by the tail kernel while it stages its rows)
            N = x.shape[0]
            state = state_out if state_out is not None else torch.empty((N, self.H), dtype=self.dtype, device=x.device)
            value = torch.empty(N, dtype=torch.float32, device=x.device)
            logits = torch.empty((N, self.A), dtype=torch.float32, device=x.device)
            self.fused_tail(y, state, value, logits, residual=x)
            return value, logits, state
        state = self._representation(obs.to(self.dtype), first)
        if state_out is not None:
            state = state_out.copy_(state)
        logits, value, _ = self._tails(self.pred1(state, relu=True), with_reward=False)
        return self._scalar(value), logits, state

    @torch.no_grad()
    def recurrent(self, hidden, action, hidden_out=None):
        state = self._dynamics(hidden, action.long(), out=hidden_out)
        logits, value, reward = self._tails(self.heads1(state, relu=True), with_reward=True)
        return self._scalar(value), self._scalar(reward), logits, state

    def support_to_scalar(self, logits):
        """inverse_value/reward_transform of [N, >=V] head outputs by the HIP kernel the fused backup uses."""
        import ctypes as C
        from ._lib import check, lib
        dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[logits.dtype]
        out = torch.empty(logits.shape[0], dtype=torch.float32, device=logits.device)
        check(lib.hz_support_to_scalar(logits.data_ptr(), logits.stride(0), self.V, -self.support, dt, out.data_ptr(),
                                       logits.shape[0], C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "hz_support_to_scalar")
        return out

    def _add_relu(self, y, res):
        """y = relu(y + res) in place: one HIP kernel (include/hz_netglue.h) on the GPU."""
        if y.device.type != "cuda":
            return torch.relu_(y.add_(res))
        import ctypes as C
        from ._lib import check, lib
        dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[y.dtype]
        assert y.stride(1) == 1 and res.stride(1) == 1 and y.shape == res.shape
        check(lib.hz_add_relu(y.data_ptr(), y.stride(0), res.data_ptr(), res.stride(0), y.shape[0], y.shape[1], dt,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hz_add_relu")
        return y

    @torch.no_grad()
    def recurrent_heads(self, net_in, hidden_out):
        """The search loop's form of recurrent_inference: `net_in` [N, H + onehot_cols] rows = [state | one-hot action
        | 0] as written by hz_tree_traverse_gather; the next hidden state goes to `hidden_out` [N, H] (its slot of the
        pool).  Returns RAW head outputs (reward_logits, value_logits, policy_logits) as strided views in the engine
        dtype, rows of width >= V / V / A -- hz_tree_backprop_nets turns them into scalars and sanitises NaNs."""
        N, H, h, hp = net_in.shape[0], self.H, self.h, self.hp
        state_in = net_in[:, :H]
        if self.full:  # NewDynamicNet
            y = self.dyn2(self.dyn1p(net_in, relu=True), relu=True)
            s = self._add_relu(self.dyn3(y, out=hidden_out), state_in)
            z = self.heads1p(s, relu=True)  # [N, 3*hp]: (reward | actor | value) blocks, each [h real | 1 | 0...]
            t2 = torch.relu_(torch.bmm(z.view(N, 3, hp).transpose(0, 1), self.bw2))
            t3 = torch.bmm(t2, self.bw3)
            ya = self._add_relu(t3[1], z[:, hp:hp + h])  # NewResMLP skip of the actor head
            return t3[0], t3[2], self.ac_tail[2](ya)
        y = self._add_relu(self.dyn1p(net_in), state_in)  # DynamicNet: skip after the first layer
        y = self.dyn2(y, relu=True)
        s = torch.relu_(self.dyn3(y, out=hidden_out))
        z = self.heads1p(s, relu=True)
        t3 = torch.bmm(z.view(N, 3, hp).transpose(0, 1), self.bw3)
        return t3[0], t3[2], t3[1]

    def flops_per_sample(self):
        """MACs*2 of one recurrent inference (for the MFMA roofline line in bench.py)."""
        mats = [self.dyn1.wt, self.dyn2.wt, self.dyn3.wt, self.heads1.wt] + [l.wt for l in self.rw_tail + self.ac_tail + self.va_tail]  # un-padded
        return 2 * sum(int(m.numel()) for m in mats)


# ------------------------------------------------------------------------------------------------ fused MFMA kernel
MLP_RELU, MLP_ACTION_ROW, MLP_BARRIER, MLP_STORE_HIDDEN, MLP_SIGNAL, MLP_BLOCKWISE, MLP_WAITS, MLP_LAST = 1, 2, 4, 8, 16, 32, 64, 128  # include/hz_mlp.h flags
MLP_F32_OUT = 2048


_FRAG_INDEX = {}


def _pack_fragments(wblk, ks, tiles=4):
    """W [<=16*tiles, <=32*ks] fp32 -> [ks][tiles][64 lanes][8] (the A-operand fragments of
    v_mfma_f32_16x16x32_bf16: lane l of tile t at k-step s holds W[16t + (l & 15)][32s + 8(l >> 4) + j], j = 0..7),
    flattened.  Runs where `wblk` lives (the host at the first build, the GPU when a learner's weights are taken over in place:
    InferenceEngine.load); the gather index of a (ks, tiles) shape is built once per device."""
    dev = wblk.device
    key = (ks, tiles, dev)
    flat = _FRAG_INDEX.get(key)
    if flat is None:
        s = torch.arange(ks).view(ks, 1, 1, 1)
        t = torch.arange(tiles).view(1, tiles, 1, 1)
        lane = torch.arange(64).view(1, 1, 64, 1)
        j = torch.arange(8).view(1, 1, 1, 8)
        n = (16 * t + (lane & 15)).expand(ks, tiles, 64, 8)
        k = (32 * s + 8 * (lane >> 4) + j).expand(ks, tiles, 64, 8)
        flat = _FRAG_INDEX[key] = (n * (32 * ks) + k).reshape(-1).to(dev)
    if tuple(wblk.shape) == (16 * tiles, 32 * ks):
        return wblk.reshape(-1)[flat] if wblk.is_contiguous() else wblk.contiguous().view(-1)[flat]
    Wp = torch.zeros(16 * tiles, 32 * ks, device=dev)
    Wp[:wblk.shape[0], :wblk.shape[1]] = wblk
    return Wp.view(-1)[flat]


class _FusedChain:
    """A chain of Linear(+folded BN)(+residual)(+ReLU) layers as the job table + per-wave weight streams of the fused
    MFMA kernel (include/hz_mlp.h).  Subclasses describe the layers with add_dense / add_group and call _finish."""

    def __init__(self, engine, waves, tiles, host_only=False, values_only=False):
        """waves x tiles: the workgroup shape of the kernel -- `waves` wavefronts, each producing `tiles` 16-column
        MFMA tiles per job (4 x 4 stand-alone, 16 x 2 inside the persistent search kernel).  Same arithmetic per
        output column either way (k accumulates in the same order), so the shapes give identical bits.
        host_only: build the tables without moving them to the engine's device (the source of another chain's reload()).
        values_only (implies host_only): only the NUMBERS -- weight streams, biases, action table -- wherever the module's
        parameters live (on the GPU for a learner's module: nothing crosses PCIe); the job table and its synchronisation plan
        (mlp_sync: two thirds of a full build's host time) depend on the layer shapes alone and are the built chain's."""
        # split: the fp16-pair build (include/hz_mlp.h, HZ_F16X2) of an fp32 engine -- every number as hi + lo halves
        self.split = bool(getattr(engine, "split", False))
        assert self.split or engine.dtype in (torch.bfloat16, torch.float16), "the fused kernel computes in bf16 or fp16 (fp32 accumulate)"
        assert not self.split or engine.dtype == torch.float32, "the fp16-pair build belongs to an fp32 engine"
        assert (waves, tiles) in ((4, 4), (8, 4), (16, 2))
        self.wdtype = torch.float16 if self.split else engine.dtype   # element format of the weight streams
        self.engine, self.device, self.waves, self.tiles = engine, engine.device, waves, tiles
        self.host_only, self.values_only = host_only or values_only, values_only
        self.cw = 16 * tiles   # output columns of one job
        self._jobs = []        # [pass] -> dict(entries [wave] -> dict or None, barrier, store_hidden)
        # (arrival counters instead of barriers: the hand-scheduled k-loop's; the fp16-pair build runs the compiler-scheduled one)
        self.blockwise = (waves, tiles) == (16, 2) and not self.split and os.environ.get("HANABIZERO_MLP_BLOCKWISE", "1") != "0"

    def _fragments(self, w, ks):
        """A job's weight block as the fragments of its wave's stream (fp32 values that the stream's format holds exactly)."""
        x = _pack_fragments(w, ks, self.tiles)
        if not self.split:
            return x
        x = x.view(ks, self.tiles, 512)
        hi = x.to(torch.float16).float()
        lo = (x - hi).to(torch.float16).float()
        return torch.stack((hi, lo), 2).reshape(-1)   # [k-step][tile][hi | lo][64 lanes x 8]

    def add_dense(self, w, b, K, src_off, dst_off, relu, res_off=None, barrier=True, store_hidden=False, act_w=None, f32=False):
        """out[:, dst_off:dst_off+nout] = act(in[:, src_off:src_off+K] @ w^T + b ...), nout split into cw-column
        wave jobs, `waves` per pass.
        16 x 2 shape: where a full-width layer (one pass, every wave 32 columns of a 512-column output) feeds the next
        layer's 512 inputs, the boundary between them is not a workgroup barrier but include/hz_mlp.h's per-block arrival
        counters (HZ_MLP_SIGNAL on the producer, HZ_MLP_BLOCKWISE on the consumer): decided here, from the chain alone."""
        cw, waves = self.cw, self.waves
        nout = w.shape[0]
        prev = self._jobs[-1] if self._jobs else None
        blockwise = bool(self.blockwise and barrier and not store_hidden and prev is not None and prev.get("full") is not None
                         and prev["full"] == src_off and K == cw * waves and len(self._jobs) - 1 < 16)
        chunks = [(c, min(cw, nout - c)) for c in range(0, nout, cw)]
        for p0 in range(0, len(chunks), waves):
            row = []
            for wave in range(waves):
                if p0 + wave >= len(chunks):
                    row.append(None)
                    continue
                c, n = chunks[p0 + wave]
                row.append(dict(w=w[c:c + n], b=b[c:c + n], ks=K // 32, src=src_off, dst=dst_off + (2 * c if f32 else c),
                                res=None if res_off is None else res_off + c, relu=relu, f32=f32,
                                act=None if act_w is None else act_w[c:c + n]))
            self._jobs.append(dict(entries=row, barrier=barrier and p0 == 0 and not blockwise, store_hidden=store_hidden and p0 == 0,
                                   blockwise=blockwise and p0 == 0,
                                   full=dst_off if (nout == cw * waves and waves == 16) else None))
        if blockwise:
            prev["signal"] = True

    def add_group(self, items, K, barrier=True, store_hidden=False):
        """independent small layers (w, b, src, dst, res, relu[, f32]) side by side: cut into cw-column jobs, `waves`
        per pass.  f32: the layer's outputs stay fp32 (HZ_MLP_F32_OUT: output column c at image columns dst + 2 c)."""
        cw, waves = self.cw, self.waves
        cut = []
        for w, b, src, dst, res, relu, *opt in items:
            f32 = bool(opt and opt[0])
            for c in range(0, w.shape[0], cw):
                cut.append(dict(w=w[c:c + cw], b=b[c:c + cw], ks=K // 32, src=src, dst=dst + (2 * c if f32 else c),
                                res=None if res is None else res + c, relu=relu, act=None, f32=f32))
        for p0 in range(0, len(cut), waves):
            row = cut[p0:p0 + waves] + [None] * (waves - len(cut[p0:p0 + waves]))
            self._jobs.append(dict(entries=row, barrier=barrier and p0 == 0, store_hidden=store_hidden and p0 == 0))

    def add_stage(self, lanes, barrier=True, store_hidden=False):
        """Independent chains side by side between two workgroup barriers: `lanes` = [[(w, b, K, src, dst, relu, res), ...],
        ...]; lane i owns the waves [i * waves / len(lanes), (i + 1) * waves / len(lanes)) and runs its layers one after
        the other on them while the other lanes stream their own: no wave idles through a barrier because the layer at
        hand is narrow.  An element of a lane may also be a LIST of layers: independent of one another, they share the lane's
        waves in the same pass(es).  No layer of a stage may read what another layer of the same stage writes (there is no
        barrier inside a stage; mlp_sync.verify checks the table that comes out); layers may have different K.  A layer may carry
        an eighth element, f32 (HZ_MLP_F32_OUT); its `dst` is then either the first image column of 2 * nout contiguous ones or a
        list with the first column of every job (a head's fp32 logits in more than one dead region of the image)."""
        cw, waves = self.cw, self.waves
        cap = waves // len(lanes)
        assert cap >= 1
        rows = []
        for i, lane in enumerate(lanes):
            r = 0
            for item in lane:
                cut = []
                for (w, b, K, src, dst, relu, res, *opt) in (item if isinstance(item, list) else [item]):
                    f32 = bool(opt and opt[0])
                    for c in range(0, w.shape[0], cw):
                        n = min(cw, w.shape[0] - c)
                        d = dst[c // cw] if isinstance(dst, (list, tuple)) else dst + (2 * c if f32 else c)
                        cut.append(dict(w=w[c:c + n], b=b[c:c + n], ks=K // 32, src=src, dst=d,
                                        res=None if res is None else res + c, relu=relu, act=None, f32=f32))
                for j, e in enumerate(cut):
                    row = r + j // cap
                    while len(rows) <= row:
                        rows.append([None] * waves)
                    rows[row][i * cap + j % cap] = e
                r += (len(cut) + cap - 1) // cap
        for k, row in enumerate(rows):
            self._jobs.append(dict(entries=row, barrier=barrier and k == 0, store_hidden=store_hidden and k == 0, fixed=True))

    def _finish(self, width, in_width, hidden, state_off, hidden_off, off_r, off_v, off_p, logit_split=None, off_r2=0, off_v2=0):
        from ._lib import MlpHeader, MlpJob
        engine, waves, tiles, cw, jobs = self.engine, self.waves, self.tiles, self.cw, self._jobs
        A, V = engine.A, 2 * engine.support + 1
        streams = [[] for _ in range(waves)]
        bias_chunks = []       # cw-float chunks
        act_rows = []          # (bias chunk index, [A, cw] block)
        assert width % 8 == 0
        lo_plane = width if self.split else 0            # (split: [hi image | lo image] per row)
        rs = (2 * width if self.split else width)
        rs += (8 - rs) % 128
        # flatten: job table, per-wave weight streams, biases, action table.  A pass with fewer jobs than waves goes to
        # the waves that have streamed the least so far (the kernel is bound by the CU's weight stream: every wave
        # should carry the same share of it and none should idle through whole layers)
        table = []
        load = [0] * waves
        rows = []
        for job in jobs:
            ents = [e for e in job["entries"] if e is not None]
            if job.get("fixed"):  # add_stage: the entries sit on the waves of their lane
                row = list(job["entries"])
                for w, e in enumerate(row):
                    load[w] += e["ks"] if e is not None else 0
            else:
                order = sorted(range(waves), key=lambda w: (load[w], w))[:len(ents)]
                row = [None] * waves
                for w, e in zip(sorted(order), ents):
                    row[w] = e
                    load[w] += e["ks"]
                assert all(e["ks"] == ents[0]["ks"] for e in ents), "jobs of one pass share their K"
            rows.append(row)
        if self.values_only:
            return self._values_only(rows, A)
        # synchronisation between the passes: barriers, except -- 16 x 2 shape -- blockwise boundaries (decided in add_dense)
        # and per-job waits (decided here from the jobs' column ranges, and proven race-free: mlp_sync)
        pass_flags = [(MLP_BARRIER if job["barrier"] else 0) | (MLP_STORE_HIDDEN if job["store_hidden"] else 0) |
                      (MLP_BLOCKWISE if job.get("blockwise") else 0) for job in jobs]
        tokens = [[[] for _ in range(waves)] for _ in jobs]
        signal_passes = {ji for ji, job in enumerate(jobs) if job.get("signal")}
        from . import mlp_sync
        sj = []
        for ji, (job, row) in enumerate(zip(jobs, rows)):
            hid = [(hidden_off, hidden_off + hidden)] if job["store_hidden"] else []
            sj.append([mlp_sync.Job(reads=hid, active=False) if e is None else
                       mlp_sync.Job(reads=[(e["src"], e["src"] + 32 * e["ks"])] + hid +
                                    ([] if e["res"] is None else [(e["res"], e["res"] + cw)]),
                                    writes=[(e["dst"], e["dst"] + (2 * cw if e.get("f32") else cw))]) for e in row])
        if self.blockwise and os.environ.get("HANABIZERO_MLP_WAITS", "1") != "0" and len(jobs) <= 16 and rs - width >= 8:
            pass_flags, sig = mlp_sync.plan(pass_flags, sj, waves)
            signal_passes |= set(sig)
            tokens = [[j.tokens for j in r] for r in sj]
        else:
            mlp_sync.verify(pass_flags, sj, waves)  # (every shape's table is checked, whatever synchronises it)
        last_job = [max((ji for ji, row in enumerate(rows) if row[w] is not None), default=-1) for w in range(waves)]
        for ji, (job, row) in enumerate(zip(jobs, rows)):
            pass_ks = max(e["ks"] for e in row if e is not None)
            if job.get("signal"):
                assert all(e is not None for e in row) and [e["dst"] for e in row] == [row[0]["dst"] + cw * w for w in range(waves)], \
                    "arrival counters: wave w produces columns [32 w, 32 w + 32)"
            if ji in signal_passes:
                assert rs - width >= 8 and ji < 16, "the counters of job j live in the padding behind image row j"
            if job.get("blockwise"):
                assert jobs[ji - 1].get("signal") and all(e is None or e["ks"] == 16 for e in row)
            for wave, e in enumerate(row):
                flags = pass_flags[ji] | (MLP_SIGNAL if ji in signal_passes else 0)
                producer = ji - 1 if job.get("blockwise") else 0
                if pass_flags[ji] & MLP_WAITS:
                    from .mlp_sync import pack_tokens
                    nt, producer = pack_tokens(tokens[ji][wave])
                    flags |= nt << 8
                if e is None:
                    table.append(MlpJob(ks=0, src_off=0, dst_off=0, res_off=-1, bias_off=0, flags=flags, reserved0=pass_ks,
                                        producer=producer))
                    continue
                assert e["ks"] % 8 == 0, "K must be a multiple of 256 (8 k-steps)"
                streams[wave].append(self._fragments(e["w"], e["ks"]))
                bias_off = cw * len(bias_chunks)
                bc = torch.zeros(cw, device=e["b"].device)
                bc[:e["b"].shape[0]] = e["b"]
                bias_chunks.append(bc)
                if e["act"] is not None:
                    act_rows.append((bias_off, e["act"]))
                    flags |= MLP_ACTION_ROW
                if e["relu"]:
                    flags |= MLP_RELU
                if e.get("f32"):
                    assert e["res"] is None and e["dst"] % 8 == 0, "an fp32 output layer has no residual and starts on a 16-B boundary"
                    flags |= MLP_F32_OUT
                if self.blockwise and last_job[wave] == ji and not job.get("blockwise"):
                    flags |= MLP_LAST
                table.append(MlpJob(ks=e["ks"], src_off=e["src"], dst_off=e["dst"],
                                    res_off=-1 if e["res"] is None else e["res"], bias_off=bias_off, flags=flags,
                                    reserved0=pass_ks, producer=producer))
        biases, act_table, W, frag = self._assemble(bias_chunks, act_rows, streams, A)
        hdr = MlpHeader(n_jobs=len(jobs), row_stride=rs, hidden=hidden, state_off=state_off, hidden_off=hidden_off,
                        off_reward=off_r, off_value=off_v, off_policy=off_p, support_size=V, support_min=-engine.support,
                        num_actions=A, action_table_stride=biases.numel(), in_width=in_width,
                        dtype=3 if self.split else {torch.bfloat16: 1, torch.float16: 2}[engine.dtype],  # HZ_F16X2 / HZ_BF16 / HZ_F16 (include/hz_tree.h)
                        lo_plane=lo_plane,
                        num_waves=waves, tiles_per_wave=tiles, kstep_stride=waves * frag,
                        logit_split=256 if logit_split is None else logit_split, off_reward2=off_r2, off_value2=off_v2)
        assert hdr.logit_split % 32 == 0 and off_r % 8 == 0 and off_v % 8 == 0 and off_p % 8 == 0 and off_r2 % 8 == 0 and off_v2 % 8 == 0
        for wave in range(waves):
            hdr.wave_stream_off[wave] = wave * frag
        self.header = hdr
        self.n_jobs = len(jobs)
        buf = (MlpJob * len(table))(*table)
        self._host = dict(jobs=torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8),
                          weights=W.reshape(-1).to(self.wdtype).contiguous(),  # (round-to-nearest-even from the fp32 fold)
                          biases=biases.float().contiguous(), act_table=act_table.float().contiguous())
        if not self.host_only:
            for k, v in self._host.items():
                setattr(self, k, v.to(self.device))
            self._host = None
        self.row_stride = rs
        self.weight_bytes_per_wg = int(sum(sum(x.numel() for x in s) for s in streams) * 2)
        self._jobs = None

    def _assemble(self, bias_chunks, act_rows, streams, A):
        """(biases, action table, interleaved weight streams, fragment size) from the per-job pieces, on the pieces' device."""
        waves, tiles = self.waves, self.tiles
        biases = torch.cat(bias_chunks)
        # the additive term of every output column, by action: the accumulators of the kernel START from a row of this table
        # (row A = the bias alone: what jobs without an action row take; row a < A = bias + the action's column of the first
        # dynamics layer), so its epilogue adds nothing
        act_table = biases.unsqueeze(0).repeat(A + 1, 1)
        for off, blk in act_rows:                           # blk [n <= cw, A] = columns of the action block
            act_table[:A, off:off + blk.shape[0]] += blk.t()
        # weight streams, interleaved k-step by k-step: [k-step][wave][tiles * 512]; at any moment the waves of a
        # workgroup (all near the same k-step) read one contiguous region
        frag = tiles * 512 * (2 if self.split else 1)
        steps = [sum(x.numel() for x in st) // frag for st in streams]
        P = max(steps) + 8                                  # 8 k-steps of zeros behind each stream (ring overrun)
        W = torch.zeros(P, waves, frag, device=biases.device)
        for wave in range(waves):
            if streams[wave]:
                W[:steps[wave], wave] = torch.cat(streams[wave]).view(-1, frag)
        return biases, act_table, W, frag

    def _values_only(self, rows, A):
        """The numbers of an already laid-out chain (same walk over the jobs as _finish's table loop, nothing else)."""
        cw, tiles = self.cw, self.tiles
        streams = [[] for _ in range(self.waves)]
        bias_chunks, act_rows = [], []
        for row in rows:
            for wave, e in enumerate(row):
                if e is None:
                    continue
                streams[wave].append(self._fragments(e["w"], e["ks"]))
                bc = torch.zeros(cw, device=e["b"].device)
                bc[:e["b"].shape[0]] = e["b"]
                if e["act"] is not None:
                    act_rows.append((cw * len(bias_chunks), e["act"]))
                bias_chunks.append(bc)
        biases, act_table, W, _ = self._assemble(bias_chunks, act_rows, streams, A)
        self.n_jobs = len(rows)
        self._host = dict(weights=W.reshape(-1).to(self.wdtype), biases=biases.float(), act_table=act_table.float())
        self._jobs = None

    def reload(self, other):
        """New weights into the device tensors this chain (and any hipGraph captured over it) already uses: `other` is the
        same chain built host_only (or values_only: no job table) from the updated module."""
        assert other.host_only and (other.waves, other.tiles, other.n_jobs) == (self.waves, self.tiles, self.n_jobs)
        assert other.values_only or bytes(other.header) == bytes(self.header), "reload(): the layer chain changed shape"
        for k, v in other._host.items():
            cur = getattr(self, k)
            assert cur.shape == v.shape and cur.dtype == v.dtype, (k, tuple(cur.shape), tuple(v.shape))
            cur.copy_(v)

    def lds_bytes(self, rows_per_wg):
        return rows_per_wg * self.row_stride * 2

    def rows_per_wg(self, N):
        return 16 if N <= 16 * 256 or self.lds_bytes(32) > 160 * 1024 else 32

    def _launch(self, state_src, row_stride, ix, plane_stride, actions, hidden_out, out_reward, out_value, out_policy, N,
                rows_per_wg=None, residual=None):
        import ctypes as C
        from ._lib import check, lib
        mt = rows_per_wg or self.rows_per_wg(N)
        check(lib.hz_mlp_recurrent_res(C.byref(self.header), self.jobs.data_ptr(), self.weights.data_ptr(),
                                       self.biases.data_ptr(), self.act_table.data_ptr(), state_src.data_ptr(), row_stride,
                                       None if ix is None else ix.data_ptr(), plane_stride, actions.data_ptr(),
                                       hidden_out.data_ptr(), out_reward.data_ptr(), out_value.data_ptr(),
                                       out_policy.data_ptr(), N, mt, None if residual is None else residual.data_ptr(),
                                       0 if residual is None else residual.stride(0),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hz_mlp_recurrent_res")


class FusedRecurrent(_FusedChain):
    """The search loop's recurrent_inference as ONE hand-written MFMA kernel (include/hz_mlp.h): gathers the parent
    hidden states from the pool, runs dynamics + reward/value/policy heads with every activation in LDS, applies the
    scalar transforms, writes the next hidden state into its pool slot.

    __call__(pool [S, N, H] bf16, ix [N] i32, actions [N] i32, hidden_out [N, H] bf16, out_reward [N], out_value [N],
             out_policy [N, A])   (fp32 outputs; buffers supplied by the caller, nothing is allocated)"""

    def __init__(self, net, engine, waves=4, tiles=4, host_only=False, values_only=False):
        super().__init__(engine, waves, tiles, host_only, values_only)
        add_dense, add_group = self.add_dense, self.add_group
        H, A, h, full, V = engine.H, engine.A, engine.h, engine.full, 2 * engine.support + 1
        dyn, rw, ac, va = net._dynamics_state, net._dynamics_reward, net._prediction_actor, net._prediction_value
        X, Y0, Y1 = 0, H, 2 * H
        w1, b1 = _fold(dyn.fc1, dyn.bn1)          # [H, H + A]: state block | action block
        w1s, w1a = w1[:, :H], w1[:, H:]           # the one-hot product = a row of w1a^T added in the epilogue
        w2, b2 = _fold(dyn.fc2, dyn.bn2)
        w3, b3 = _fold(dyn.fc3, dyn.bn3)
        wh = torch.cat([_fold(rw[0], rw[1])[0], _fold(ac[0], ac[1])[0], _fold(va[0], va[1])[0]], 0)
        bh = torch.cat([_fold(rw[0], rw[1])[1], _fold(ac[0], ac[1])[1], _fold(va[0], va[1])[1]], 0)
        if full:   # NewDynamicNet + 3-layer heads (config/hanabi_control/model.py:93-125, 250-269)
            # Row image, 3 H + h wide (every region is reused as soon as its contents are dead, so that 32 rows of it and
            # the 32 trees' search state fit the 160 KiB of a workgroup -- hz_search.hip):
            #   [0, H) X state | [H, 2H) Y1 | [2H, 3H) Y0 = the next hidden state | [3H, 3H + h) E
            # After the dynamics net the three head chains (reward R, actor A, value V: h1 -> h2 -> h3 (-> policy)) are
            # independent, and a 256-column layer occupies only half the waves of a 16 x 2 workgroup: instead of one layer
            # per barrier the chains are staggered so that between two barriers both halves stream --
            #   pass   h1R | h1A                      Y0 -> Z = [0, 2h)        (all waves; stores the hidden state)
            #   S1     h1V            || h2A, h2R     Y0 -> [2h, 3h);  Z thirds -> Ta = E, Tr = [3h, 4h)
            #   S2     h2V            || h3A          [2h, 3h) -> Tv = Y0's first h;  Ta -> [h, 2h) in place (+ residual)
            #   S3     h3V            || h3R | policy Tv -> [2h, 3h);  Tr -> [0, h);  [h, 2h) -> E
            assert 3 * h <= 2 * H and 4 * h <= 2 * H and 2 * H + h <= 3 * H
            Y1, Y0, E = H, 2 * H, 3 * H
            Z = X
            wr1, br1 = _fold(rw[0], rw[1])
            wa1, ba1 = _fold(ac[0], ac[1])
            wv1, bv1 = _fold(va[0], va[1])
            add_dense(w1s, b1, H, X, Y0, relu=True, barrier=False, act_w=w1a)
            add_dense(w2, b2, H, Y0, Y1, relu=True)
            add_dense(w3, b3, H, Y1, Y0, relu=True, res_off=X)
            # (the next hidden state, Y0, leaves for the pool behind the NEXT barrier -- S1's, Y0 is intact until S2 -- so that
            # this layer's boundary, too, can be the blockwise one)
            add_dense(torch.cat([wr1, wa1], 0), torch.cat([br1, ba1], 0), H, Y0, Z, relu=True)
            Tr, Ta, Tv = Z + 3 * h, E, Y0
            wr2, br2 = _fold(rw[3], rw[4])
            wa2, ba2 = _fold(ac[3].fc1, ac[3].bn1)
            wv2, bv2 = _fold(va[3], va[4])
            wr3, br3 = _fold(rw[6])
            wa3, ba3 = _fold(ac[3].fc2, ac[3].bn2)
            wv3, bv3 = _fold(va[6])
            wp4, bp4 = _fold(ac[4])
            R3, U, V3 = Z, Z + h, Z + 2 * h                 # reward logits | actor hidden | value logits
            self.add_stage([[(wv1, bv1, H, Y0, Z + 2 * h, True, None)],
                            [(wa2, ba2, h, Z + h, Ta, True, None), (wr2, br2, h, Z, Tr, True, None)]], store_hidden=True)
            # (h2A before h2R: the actor head is the longest chain of dependent layers behind the dynamics net)
            self.add_stage([[(wv2, bv2, h, Z + 2 * h, Tv, True, None)],
                            [(wa3, ba3, h, Ta, U, True, Z + h)]])  # (in place: a lane reads the residual element it then overwrites)
            # the three output layers in ONE last pass: value logits on lane 0, reward logits and policy side by side on lane 1 (the
            # passes with a few jobs each are bound by a wave's own latency, not by the stream: h3R used to follow h2V on lane 0
            # and the policy had a pass of its own).  All three stay fp32 (HZ_MLP_F32_OUT: 2 image columns per logit) on their way
            # to the scalar transform and the tree; they land in what is dead by now -- [0, h) (h1R's output), [2h, 3h) (h1V's),
            # the second half of Y0 (the hidden state has left) and E (Ta, read by h3A) -- a head's 201 logits in two pieces:
            # logits [0, 128) and the rest (header fields logit_split / off_*2).  The 4-tile shapes (64-column jobs: their
            # last jobs are mostly padding) get 128 more columns behind the image for the policy; they run stand-alone.
            split, cw = 128, self.cw
            nj1, njV = split // cw, (V + cw - 1) // cw
            seg = lambda a, b: [a + 2 * cw * j for j in range(nj1)] + [b + 2 * cw * j for j in range(njV - nj1)]
            off_v, off_v2, off_r = 2 * h, Y0 + h, 0
            off_r2 = off_v2 + 2 * cw * (njV - nj1)
            off_p = off_r2 + 2 * cw * (njV - nj1)
            width = max(3 * H + h, off_p + 2 * cw)
            assert off_v2 >= Tv + h and off_r2 + 2 * cw * (njV - nj1) <= off_p
            self.add_stage([[(wv3, bv3, h, Tv, seg(off_v, off_v2), False, None, True)],
                            [[(wr3, br3, h, Tr, seg(off_r, off_r2), False, None, True), (wp4, bp4, h, U, off_p, False, None, True)]]])
            fin = dict(off_r=off_r, off_v=off_v, off_p=off_p, logit_split=split, off_r2=off_r2, off_v2=off_v2)
        else:      # DynamicNet + 2-layer heads (model.py:61-91, 138-149)
            Z = Y1
            add_dense(w1s, b1, H, X, Y0, relu=True, res_off=X, barrier=False, act_w=w1a)
            add_dense(w2, b2, H, Y0, Y1, relu=True)
            add_dense(w3, b3, H, Y1, Y0, relu=True)
            add_dense(wh, bh, H, Y0, Z, relu=True)
            assert V <= 64 and A <= 64 and 2 * h <= 256
            kpad = 256                                      # K = h padded to 8 k-steps with zero weights
            # (fp32 logits, HZ_MLP_F32_OUT: <= 64 logits = 128 image columns per head, over the dead input state)
            outs = [(_fold(rw[3]), Z, X), (_fold(ac[3]), Z + h, X + 256), (_fold(va[3]), Z + 2 * h, X + 128)]
            add_group([(w, b, src, dst, None, False, True) for (w, b), src, dst in outs], kpad, store_hidden=True)  # (Y0 is still intact)
            fin = dict(off_r=X, off_v=X + 128, off_p=X + 256)
            width = 3 * H
        self._finish(width, in_width=H, hidden=H, state_off=X, hidden_off=Y0, **fin)

    def __call__(self, pool, ix, actions, hidden_out, out_reward, out_value, out_policy, rows_per_wg=None):
        """pool [S, N, H] (or a [N, H] matrix of states with ix=None)."""
        if pool.dim() == 3:
            N, row_stride, plane_stride = pool.shape[1], pool.stride(1), pool.stride(0)
        else:
            N, row_stride, plane_stride = pool.shape[0], pool.stride(0), 0
        self._launch(pool, row_stride, ix, plane_stride, actions, hidden_out, out_reward, out_value, out_policy, N,
                     rows_per_wg)
        return out_reward, out_value, out_policy


class FusedInitialTail(_FusedChain):
    """The small-GEMM tail of initial_inference (core/model.py:61-71) for MuZeroNetFull as ONE launch of the same MFMA
    kernel: from the output of the first residual block of the representation net (config/hanabi_control/model.py:
    235-248: Linear-BN-ReLU 1024 -> 512, NewResMLP(512)) through the prediction net (:250-269) to the hidden state,
    the value scalar and the policy logits.  In PyTorch that is 11 GEMMs of 4096 x <= 512 x <= 1024 plus glue, each
    bound by its launch; the two large representation layers before it stay hipBLASLt GEMMs.

    __call__(x [N, 1024] bf16, hidden_out [N, H] bf16, out_value [N] f32, out_policy [N, A] f32)"""

    def __init__(self, net, engine, waves=4, tiles=4, host_only=False, values_only=False):
        super().__init__(engine, waves, tiles, host_only, values_only)
        assert engine.full, "laid out for MuZeroNetFull"
        add_dense, add_group = self.add_dense, self.add_group
        H, A, h, V = engine.H, engine.A, engine.h, 2 * engine.support + 1
        rep, ac, va = net._representation, net._prediction_actor, net._prediction_value
        w3, b3 = _fold(rep[4], rep[5])                     # 1024 -> H
        IN = w3.shape[1]
        assert IN % 256 == 0 and H % 256 == 0 and h % 256 == 0 and IN >= 2 * H and 2 * h <= H
        T0 = IN                                             # image columns: IN [0, IN) | T0 [IN, IN + H) | U | V3
        T1, S, Z = 0, H, T0                                 # T1 and S reuse the dead input, Z reuses T0
        Ta, Tv = 0, h                                       # ... and the second head layers reuse T1
        U, V3 = IN + H, IN + H + h
        add_dense(w3, b3, IN, 0, T0, relu=True, barrier=False)
        add_dense(*_fold(rep[7].fc1, rep[7].bn1), H, T0, T1, relu=True)
        add_dense(*_fold(rep[7].fc2, rep[7].bn2), H, T1, S, relu=True, res_off=T0)
        wh = torch.cat([_fold(ac[0], ac[1])[0], _fold(va[0], va[1])[0]], 0)
        bh = torch.cat([_fold(ac[0], ac[1])[1], _fold(va[0], va[1])[1]], 0)
        add_dense(wh, bh, H, S, Z, relu=True, store_hidden=True)           # actor | value first layers
        add_group([(*_fold(ac[3].fc1, ac[3].bn1), Z, Ta, None, True), (*_fold(va[3], va[4]), Z + h, Tv, None, True)], h)
        add_group([(*_fold(ac[3].fc2, ac[3].bn2), Ta, U, Z, True), (*_fold(va[6]), Tv, V3, None, False, True)], h)  # (value logits: fp32)
        add_dense(*_fold(ac[4]), h, U, Z, relu=False, barrier=True, f32=True)  # policy logits (fp32) over the dead Z
        width = V3 + 2 * ((V + 63) // 64) * 64
        self.in_width = IN
        self._finish(width, in_width=IN, hidden=H, state_off=0, hidden_off=S, off_r=V3, off_v=V3, off_p=Z)
        self._zeros = None

    def __call__(self, x, hidden_out, out_value, out_policy, rows_per_wg=None, residual=None):
        """residual: the chain's input rows are relu(x + residual) (the skip of the representation net's first residual block,
        added while the rows are staged: include/hz_mlp.h::hz_mlp_recurrent_res)."""
        N = x.shape[0]
        assert x.dtype == self.engine.dtype and x.shape[1] == self.in_width and x.stride(1) == 1
        assert residual is None or (residual.dtype == x.dtype and residual.shape == x.shape and residual.stride(1) == 1)
        if self._zeros is None or self._zeros[0].shape[0] < N:
            self._zeros = (torch.zeros(N, dtype=torch.int32, device=x.device), torch.empty(N, dtype=torch.float32, device=x.device))
        actions, dummy = self._zeros
        self._launch(x, x.stride(0), None, 0, actions, hidden_out, dummy, out_value, out_policy, N, rows_per_wg, residual=residual)
        return out_value, out_policy


