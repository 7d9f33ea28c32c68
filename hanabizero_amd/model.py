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
    """y = x @ Wt + b with Wt [in, out] resident in `dtype`; optional in-place ReLU."""

    def __init__(self, w, b, dtype, device):
        self.wt = w.t().contiguous().to(device=device, dtype=dtype)
        self.b = b.to(device=device, dtype=dtype)

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
        include/hz_mlp.h (default: whenever the engine is bf16 on a GPU); False keeps the hipBLASLt GEMM chain."""
        self.dtype, self.device = dtype, torch.device(device)
        self.A, self.H = net.action_space_n, net.feature_size
        self.support = int(support)
        self.full = isinstance(net, MuZeroNetFull)
        self.use_fused = (dtype == torch.bfloat16 and self.device.type == "cuda") if fused is None else bool(fused)
        self.fused = None
        self.load(net)

    def load(self, net):
        """(Re)build the folded weights from `net` (call after set_weights: selfplay_worker.py:177-184)."""
        L = lambda lin, bn=None: _Lin(*_fold(lin, bn), self.dtype, self.device)
        rep, dyn = net._representation, net._dynamics_state
        rw, ac, va = net._dynamics_reward, net._prediction_actor, net._prediction_value
        if self.full:
            self.rep = [L(rep[0], rep[1]), L(rep[3].fc1, rep[3].bn1), L(rep[3].fc2, rep[3].bn2), L(rep[4], rep[5]),
                        L(rep[7].fc1, rep[7].bn1), L(rep[7].fc2, rep[7].bn2)]
        else:
            self.rep = [L(rep[0], rep[1]), L(rep[3].fc1, rep[3].bn1), L(rep[3].fc2, rep[3].bn2)]
        # dynamics layer 1: split [W_state | W_action]; the one-hot product is a row lookup of W_action^T
        w1, b1 = _fold(dyn.fc1, dyn.bn1)
        self.dyn1 = _Lin(w1[:, :self.H], b1, self.dtype, self.device)
        self.dyn1_act = w1[:, self.H:].t().contiguous().to(device=self.device, dtype=self.dtype)  # [A, H]
        # ... and the same layer over the [state | one-hot | 0-pad] rows the traverse kernel writes (K padded to 32)
        self.onehot_cols = ((self.A + 31) // 32) * 32
        w1p = torch.zeros(w1.shape[0], self.H + self.onehot_cols)
        w1p[:, :w1.shape[1]] = w1
        self.dyn1p = _Lin(w1p, b1, self.dtype, self.device)
        self.dyn2, self.dyn3 = L(dyn.fc2, dyn.bn2), L(dyn.fc3, dyn.bn3)
        # the three heads' first layers share their input: one GEMM [512 -> 3h] (reward | actor | value)
        ws, bs = zip(_fold(rw[0], rw[1]), _fold(ac[0], ac[1]), _fold(va[0], va[1]))
        self.h = ws[0].shape[0]
        self.heads1 = _Lin(torch.cat(ws, 0), torch.cat(bs, 0), self.dtype, self.device)
        self.pred1 = _Lin(torch.cat(ws[1:], 0), torch.cat(bs[1:], 0), self.dtype, self.device)
        # search-loop form: each head block is h real columns + 32 pad columns whose first one is a constant 1
        # (zero weights, bias 1, ReLU(1) = 1): it carries the next layers' biases through plain batched GEMMs
        self.hp = self.h + 32
        wpad, bpad = [], []
        for w, b in zip(ws, bs):
            wpad.append(torch.cat((w, torch.zeros(32, w.shape[1])), 0))
            one = torch.zeros(32)
            one[0] = 1.0
            bpad.append(torch.cat((b, one), 0))
        self.heads1p = _Lin(torch.cat(wpad, 0), torch.cat(bpad, 0), self.dtype, self.device)
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
        self.fused = FusedRecurrent(net, self) if self.use_fused else None

    def _stack(self, folded, out_width, carry_one):
        """[(W [o, h], b [o])] * 3 -> Wt [3, hp, out_width] in the engine dtype for torch.bmm over the three head
        blocks: rows 0..h-1 = W^T, row h = b (multiplied by the constant-one input column), other rows 0; with
        carry_one the output keeps a constant-one column at index h for the next layer's bias."""
        out = torch.zeros(3, self.hp, out_width)
        for k, (w, b) in enumerate(folded):
            out[k, :w.shape[1], :w.shape[0]] = w.t()
            out[k, self.h, :b.shape[0]] = b
            if carry_one:
                out[k, self.h, self.h] = 1.0
        return out.to(device=self.device, dtype=self.dtype)

    # -- pieces ---------------------------------------------------------------------------------------
    def _scalar(self, logits):
        return inverse_scalar_transform(logits, -self.support, self.support).reshape(-1)

    def _representation(self, x):
        r = self.rep
        if self.full:  # Linear-BN-ReLU, NewResMLP(1024), Linear-BN-ReLU, NewResMLP(512)
            x = r[0](x, relu=True)
            x = torch.relu_(r[2](r[1](x, relu=True)) + x)
            x = r[3](x, relu=True)
            return torch.relu_(r[5](r[4](x, relu=True)) + x)
        x = r[0](x, relu=True)  # Linear-BN-ReLU, ResMLP(512)
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
            ya = torch.relu_(a[1](a[0](za, relu=True)) + za)
            logits = a[2](ya)
            value = self.va_tail[1](self.va_tail[0](zv, relu=True))
            reward = self.rw_tail[1](self.rw_tail[0](z[:, :h], relu=True)) if with_reward else None
        else:
            logits, value = self.ac_tail[0](za), self.va_tail[0](zv)
            reward = self.rw_tail[0](z[:, :h]) if with_reward else None
        return logits.float(), value, reward

    # -- entry points -----------------------------------------------------------------------------------
    @torch.no_grad()
    def initial(self, obs):
        state = self._representation(obs.to(self.dtype))
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
_MLP_KINDS = {(8, 1, 17): 0, (8, 1, 16): 1, (12, 1, 16): 2, (12, 3, 8): 3, (1, 1, 8): 4, (6, 1, 16): 5, (3, 3, 4): 6,
              (8, 1, 18): 7}  # (tiles per wave, groups, k-steps) instantiated in csrc/hz_mlp.hip


def _pack_layer(groups_wb, K, nout_pad_per_group):
    """[(W [o, K], b [o])] per group -> (packed bf16-to-be weights as a flat fp32 tensor, bias fp32 [G * nout_pad]).
    Packed order [wave 0..3][k-step][tile][lane 0..63][8]: lane l of tile t holds W[n = 16*tile + (l & 15)]
    [k = 32*s + 8*(l >> 4) + j], the A-operand fragment of v_mfma_f32_16x16x32_bf16 (csrc/hz_mlp.hip)."""
    G = len(groups_wb)
    TG = nout_pad_per_group // 64  # tiles per wave per group
    TW, KS = TG * G, K // 32
    Wp = torch.zeros(G, nout_pad_per_group, K)
    bias = torch.zeros(G, nout_pad_per_group)
    for g, (w, b) in enumerate(groups_wb):
        assert w.shape[1] <= K
        Wp[g, :w.shape[0], :w.shape[1]] = w
        bias[g, :b.shape[0]] = b
    wave = torch.arange(4).view(4, 1, 1, 1, 1)
    s = torch.arange(KS).view(1, KS, 1, 1, 1)
    t = torch.arange(TW).view(1, 1, TW, 1, 1)
    lane = torch.arange(64).view(1, 1, 1, 64, 1)
    j = torch.arange(8).view(1, 1, 1, 1, 8)
    g = t // TG
    n = 16 * (wave * TG + t % TG) + (lane & 15)
    k = 32 * s + 8 * (lane >> 4) + j
    packed = Wp[g.expand(4, KS, TW, 64, 8), n.expand(4, KS, TW, 64, 8), k.expand(4, KS, TW, 64, 8)]
    return packed.reshape(-1), bias.reshape(-1), _MLP_KINDS[(TW, G, KS)]


class FusedRecurrent:
    """recurrent_inference of an InferenceEngine's network as ONE hand-written MFMA kernel (include/hz_mlp.h).
    __call__(net_in [N, H + onehot_cols] bf16, hidden_out [N, H] bf16) -> (reward [N], value [N], policy [N, A]) f32."""

    def __init__(self, net, engine):
        from ._lib import MlpLayer, MlpProgram
        assert engine.dtype == torch.bfloat16, "the fused kernel computes in bf16 (fp32 accumulate)"
        self.engine, self.device = engine, engine.device
        H, A, h, full = engine.H, engine.A, engine.h, engine.full
        oh = engine.onehot_cols
        KX = H + oh
        dyn = net._dynamics_state
        rw, ac, va = net._dynamics_reward, net._prediction_actor, net._prediction_value
        V = engine.V
        layers, packed, biases = [], [], []

        def add(groups_wb, K, nout_g, src_off, dst_off, src_gstride=0, res_off=-1, res_group=-1, relu_mask=0, store_hidden=0):
            w, b, kind = _pack_layer(groups_wb, K, nout_g)
            L = MlpLayer(K=K, nout=nout_g * len(groups_wb), groups=len(groups_wb), src_off=src_off, src_gstride=src_gstride,
                         dst_off=dst_off, res_off=res_off, res_group=res_group, relu_mask=relu_mask,
                         store_hidden=store_hidden, w_off=sum(x.numel() for x in packed),
                         b_off=sum(x.numel() for x in biases), kind=kind)
            layers.append(L)
            packed.append(w)
            biases.append(b)

        X, Y1, Y0 = 0, KX, KX + H          # LDS columns: [state|one-hot] , second dynamics buffer, first dynamics buffer
        Z = Y0 + H                          # head hidden block(s)
        w1, b1 = _fold(dyn.fc1, dyn.bn1)    # [H, H + A] -> K padded to KX with zero columns
        if full:
            add([(w1, b1)], KX, H, X, Y0, relu_mask=1)
            add([_fold(dyn.fc2, dyn.bn2)], H, H, Y0, Y1, relu_mask=1)
            add([_fold(dyn.fc3, dyn.bn3)], H, H, Y1, Y0, res_off=X, relu_mask=1, store_hidden=1)
            add([(torch.cat([_fold(rw[0], rw[1])[0], _fold(ac[0], ac[1])[0], _fold(va[0], va[1])[0]], 0),
                  torch.cat([_fold(rw[0], rw[1])[1], _fold(ac[0], ac[1])[1], _fold(va[0], va[1])[1]], 0))],
                H, 3 * h, Y0, Z, relu_mask=1)
            T = 0                           # X and Y1 are dead: second head layer [T, T + 3h)
            add([_fold(rw[3], rw[4]), _fold(ac[3].fc1, ac[3].bn1), _fold(va[3], va[4])], h, h, Z, T, src_gstride=h,
                relu_mask=0b111)
            O = Y0                          # third head layer over Y0 and the dead reward block of Z
            # the actor group (layer columns h..2h) adds Z's actor block: residual column = res_off + layer column
            add([_fold(rw[6]), _fold(ac[3].fc2, ac[3].bn2), _fold(va[6])], h, h, T, O, src_gstride=h,
                res_off=Z, res_group=1, relu_mask=0b010)
            add([_fold(ac[4])], h, 64, O + h, 0)
            off_r, off_v, off_p = O, O + 2 * h, 0
            width = Z + 3 * h
        else:
            add([(w1, b1)], KX, H, X, Y0, res_off=X, relu_mask=1)
            add([_fold(dyn.fc2, dyn.bn2)], H, H, Y0, Y1, relu_mask=1)
            add([_fold(dyn.fc3, dyn.bn3)], H, H, Y1, Y0, relu_mask=1, store_hidden=1)
            Z = Y1                          # Y1 is dead after the third dynamics layer
            add([(torch.cat([_fold(rw[0], rw[1])[0], _fold(ac[0], ac[1])[0], _fold(va[0], va[1])[0]], 0),
                  torch.cat([_fold(rw[0], rw[1])[1], _fold(ac[0], ac[1])[1], _fold(va[0], va[1])[1]], 0))],
                H, 3 * h, Y0, Z, relu_mask=1)
            assert V <= 64 and A <= 64
            add([_fold(rw[3]), _fold(ac[3]), _fold(va[3])], h, 64, Z, 0, src_gstride=h)
            off_r, off_p, off_v = 0, 64, 128
            width = Y0 + H
        rs = width + ((8 - width) % 128)    # row stride = 8 (mod 128) elements: conflict-free ds_read_b128 over 16 rows
        P = MlpProgram(n_layers=len(layers), row_stride=rs, in_width=KX, hidden=H, off_reward=off_r, off_value=off_v,
                       off_policy=off_p, support_size=V, support_min=-engine.support, num_actions=A)
        for i, L in enumerate(layers):
            P.layer[i] = L
        self.program = P
        self.weights = torch.cat(packed).to(device=self.device, dtype=torch.bfloat16).contiguous()
        self.biases = torch.cat(biases).to(device=self.device, dtype=torch.float32).contiguous()
        self.rows_per_wg = 16
        self.lds_bytes = lambda mt: mt * rs * 2

    def __call__(self, net_in, hidden_out, out_reward, out_value, out_policy):
        import ctypes as C
        from ._lib import check, lib
        N = net_in.shape[0]
        mt = 16 if N <= 16 * 256 or self.lds_bytes(32) > 160 * 1024 else 32
        check(lib.hz_mlp_recurrent(C.byref(self.program), net_in.data_ptr(), net_in.stride(0), self.weights.data_ptr(),
                                   self.biases.data_ptr(), hidden_out.data_ptr(), out_reward.data_ptr(),
                                   out_value.data_ptr(), out_policy.data_ptr(), N, mt,
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hz_mlp_recurrent")
        return out_reward, out_value, out_policy
