"""hanabizero_amd.cytree -- drop-in for the reference's Cython module ``core.ctree.cytree``.

Same names, argument order and return types as /root/reference/core/ctree/cytree.pyx:
    Roots(root_num, action_num, tree_nodes)   .prepare / .prepare_no_noise / .get_trajectories /
                                              .get_distributions / .get_values / .clear / .num     (:37-70)
    MinMaxStatsList(num).set_delta(d)                                                              (:17-27)
    ResultsWrapper(num)                                                                            (:30-34)
    multi_traverse(roots, pb_c_base, pb_c_init, discount, min_max_stats_lst, results)              (:97-101)
    multi_back_propagate(hidden_state_index_x, discount, rewards, values, policies, mm, results)   (:87-94)
    Node                                                                    (imported, never used: :73-85)

The tree itself lives in HBM and is searched by the HIP kernels of libhanabizero_hip.so (include/hz_tree.h).
List arguments are accepted exactly as the reference takes them (compatibility path: one H2D copy each);
torch CUDA tensors are accepted everywhere a list is, and the ``*_tensors`` methods keep results on the device
(fast path used by hanabizero_amd.mcts).  There is no CPU implementation behind this module.

Differences a caller can observe (documented in DESIGN.md):
  * ties between near-equal pUCT scores are broken by the counter-based stream of include/hz_tiebreak.h
    (seed = ``Roots.tie_seed``) instead of libc rand() reseeded from the wall clock (cnode.cpp:409-411);
  * hidden_state_index_x passed to multi_back_propagate must advance 1, 2, 3, ... after prepare, which is what
    core/mcts.py:53-57 does.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib

_pool = {}  # (N, A, S, device) -> [free hz_tree_t handles]; Roots objects are built once per move in the reference


def _device_index(device):
    if device is None:
        return torch.cuda.current_device()
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(x, dtype, shape, device):
    """lists / numpy / tensors -> contiguous CUDA tensor of `dtype` and `shape`."""
    if isinstance(x, torch.Tensor):
        t = x
        if t.device.type != "cuda":
            t = t.to(device, non_blocking=True)
        if t.dtype != dtype:
            t = t.to(dtype)
    else:
        t = torch.as_tensor(np.asarray(x), device=device).to(dtype)
    t = t.reshape(shape)
    return t if t.is_contiguous() else t.contiguous()


class MinMaxStatsList:
    """tools::CMinMaxStatsList (cminimax.cpp:48-63); the statistics themselves live in the tree handle."""

    def __init__(self, num):
        self.num = int(num)
        self.value_delta_max = 0.0

    def set_delta(self, value_delta_max):
        self.value_delta_max = float(value_delta_max)


class ResultsWrapper:
    """tree::CSearchResults (cnode.cpp:6-17); the search paths live in the tree handle between
    multi_traverse and multi_back_propagate of the same simulation."""

    def __init__(self, num):
        self.num = int(num)
        self.roots = None


class Node:
    """Placeholder kept because core/game.py:4 imports it; the reference never instantiates it usefully."""

    def __init__(self, prior=0.0, action_num=0):
        self.prior, self.action_num = prior, action_num


class Roots:
    def __init__(self, root_num, action_num, tree_nodes, device=None, tie_seed=0, tree_id_base=0):
        self.root_num, self.action_num, self.tree_nodes = int(root_num), int(action_num), int(tree_nodes)
        self.pool_size = self.action_num * (self.tree_nodes + 2)  # cytree.pyx:44 (informational)
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        self.tie_seed, self.tree_id_base = int(tie_seed), int(tree_id_base)
        self._key = (self.root_num, self.action_num, self.tree_nodes, self.device_index)
        free = _pool.setdefault(self._key, [])
        if free:
            self._h = free.pop()
        else:
            h = C.c_void_p()
            check(lib.hz_tree_create(C.byref(h), self.root_num, self.action_num, self.tree_nodes, self.device_index),
                  "hz_tree_create")
            self._h = h
        self._params = None
        self._sim = 0
        N = self.root_num
        self._ix = torch.empty(N, dtype=torch.int32, device=self.device)
        self._iy = torch.empty(N, dtype=torch.int32, device=self.device)
        self._la = torch.empty(N, dtype=torch.int32, device=self.device)

    # -- lifetime -------------------------------------------------------------------------------
    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and _pool is not None:
            _pool.setdefault(self._key, []).append(h)
            self._h = None

    def clear(self):  # CRoots::clear (cnode.cpp:261-264)
        self.__del__()

    @property
    def num(self):
        return self.root_num

    @property
    def hbm_bytes(self):
        return int(lib.hz_tree_hbm_bytes(self._h))

    def clone(self):
        """A second Roots holding a device-side copy of this one's complete search state (hz_tree_copy)."""
        other = Roots(self.root_num, self.action_num, self.tree_nodes, device=self.device, tie_seed=self.tie_seed,
                      tree_id_base=self.tree_id_base)
        check(lib.hz_tree_copy(other._h, self._h, _stream()), "hz_tree_copy")
        other._params, other._sim = self._params, self._sim
        return other

    # -- parameters the reference passes per call --------------------------------------------------
    def set_params(self, pb_c_base, pb_c_init, discount, value_delta_max):
        p = (int(pb_c_base), float(pb_c_init), float(discount), float(value_delta_max), self.tie_seed,
             self.tree_id_base)
        if p != self._params:
            check(lib.hz_tree_set_params(self._h, *p), "hz_tree_set_params")
            self._params = p

    # -- prepare ---------------------------------------------------------------------------------------
    def prepare(self, root_exploration_fraction, noises, reward_pool, policy_logits_pool, stack_legal_action):
        N, A, dev = self.root_num, self.action_num, self.device
        nz = _dev(noises, torch.float32, (N, A), dev)
        rw = _dev(reward_pool, torch.float32, (N,), dev)
        lg = _dev(policy_logits_pool, torch.float32, (N, A), dev)
        la = _dev(stack_legal_action, torch.uint8, (N, A), dev)
        check(lib.hz_tree_prepare(self._h, float(root_exploration_fraction), nz.data_ptr(), rw.data_ptr(),
                                  lg.data_ptr(), la.data_ptr(), _stream()), "hz_tree_prepare")
        self._keep = (nz, rw, lg, la)
        self._sim = 0

    def prepare_no_noise(self, reward_pool, policy_logits_pool, stack_legal_action):
        N, A, dev = self.root_num, self.action_num, self.device
        rw = _dev(reward_pool, torch.float32, (N,), dev)
        lg = _dev(policy_logits_pool, torch.float32, (N, A), dev)
        la = _dev(stack_legal_action, torch.uint8, (N, A), dev)
        check(lib.hz_tree_prepare(self._h, 0.0, None, rw.data_ptr(), lg.data_ptr(), la.data_ptr(), _stream()),
              "hz_tree_prepare")
        self._keep = (rw, lg, la)
        self._sim = 0

    # -- device-resident fast path ------------------------------------------------------------------
    def traverse_tensors(self, pool=None, net_in=None, onehot_cols=0):
        """One descent per tree.  Returns (ix, iy, last_action) int32 CUDA tensors (reused buffers).
        With `pool` [S, N, H] and `net_in` [N, >=H]: also gathers pool[ix, tree] into net_in[:, :H] and, with
        onehot_cols > 0, writes one_hot(last_action) into net_in[:, H:H+onehot_cols]."""
        if pool is None:
            check(lib.hz_tree_traverse(self._h, self._sim, self._ix.data_ptr(), self._iy.data_ptr(),
                                       self._la.data_ptr(), _stream()), "hz_tree_traverse")
        else:
            dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[pool.dtype]
            assert pool.is_contiguous() and pool.shape[1] == self.root_num and net_in.dtype == pool.dtype
            assert net_in.stride(1) == 1
            check(lib.hz_tree_traverse_gather(self._h, self._sim, self._ix.data_ptr(), self._iy.data_ptr(),
                                              self._la.data_ptr(), pool.data_ptr(), pool.shape[2], dt,
                                              net_in.data_ptr(), net_in.stride(0), int(onehot_cols), _stream()),
                  "hz_tree_traverse_gather")
        self._sim += 1
        return self._ix, self._iy, self._la

    def backprop_tensors(self, hidden_state_index_x, rewards, values, policy_logits):
        N, A, dev = self.root_num, self.action_num, self.device
        rw = _dev(rewards, torch.float32, (N,), dev)
        vl = _dev(values, torch.float32, (N,), dev)
        lg = _dev(policy_logits, torch.float32, (N, A), dev)
        check(lib.hz_tree_backprop(self._h, int(hidden_state_index_x), rw.data_ptr(), vl.data_ptr(), lg.data_ptr(),
                                   _stream()), "hz_tree_backprop")
        self._keep_bp = (rw, vl, lg)

    def backprop_traverse_tensors(self, hidden_state_index_x, rewards, values, policy_logits):
        """multi_back_propagate of this simulation + multi_traverse of the next one in ONE launch (fp32 CUDA tensors).
        Returns (ix, iy, last_action) of the next descent."""
        check(lib.hz_tree_backprop_traverse(self._h, int(hidden_state_index_x), rewards.data_ptr(), values.data_ptr(),
                                            policy_logits.data_ptr(), self._sim, self._ix.data_ptr(),
                                            self._iy.data_ptr(), self._la.data_ptr(), _stream()),
              "hz_tree_backprop_traverse")
        self._sim += 1
        return self._ix, self._iy, self._la

    def backprop_nets_tensors(self, hidden_state_index_x, reward_logits, value_logits, support_size, support_min,
                              policy_logits, out_rewards=None, out_values=None):
        """multi_back_propagate fed by raw head outputs (strided 2-D CUDA tensors of one dtype): include/hz_tree.h
        hz_tree_backprop_nets."""
        dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[value_logits.dtype]
        for t in (reward_logits, value_logits, policy_logits):
            assert t.dtype == value_logits.dtype and t.stride(1) == 1 and t.shape[0] == self.root_num
        check(lib.hz_tree_backprop_nets(self._h, int(hidden_state_index_x), reward_logits.data_ptr(),
                                        reward_logits.stride(0), value_logits.data_ptr(), value_logits.stride(0),
                                        int(support_size), int(support_min), policy_logits.data_ptr(),
                                        policy_logits.stride(0), dt,
                                        None if out_rewards is None else out_rewards.data_ptr(),
                                        None if out_values is None else out_values.data_ptr(), _stream()),
              "hz_tree_backprop_nets")

    def distributions_tensor(self):
        out = torch.empty((self.root_num, self.action_num), dtype=torch.int32, device=self.device)
        check(lib.hz_tree_get_distributions(self._h, out.data_ptr(), _stream()), "hz_tree_get_distributions")
        return out

    def search_tensors(self, fused, pool, num_simulations, rew, val, pol, rows_per_workgroup=0):
        """Every simulation of one move in ONE persistent kernel (include/hz_search.h): `fused` is the engine's
        FusedRecurrent laid out for 16 waves x 2 tiles, `pool` [S, N, H] in the engine's bf16 / fp16 with the root states
        in plane 0; rew / val [N] f32 and pol [N, A] f32 are scratch; rows_per_workgroup: 0 = auto (16 | 32 | -32 force a
        kernel shape: tests, measurements).  Returns (ix, iy, last_actions) of the last simulation."""
        import ctypes as C
        assert fused.waves == 16 and fused.tiles == 2 and pool.dim() == 3 and pool.shape[1] == self.root_num
        assert pool.dtype == fused.engine.dtype, "the pool must be in the fused MLP's element format"
        assert self._sim == 0, "search_tensors needs freshly prepared roots"
        ix, iy, la = self._ix, self._iy, self._la
        check(lib.hz_search_run(self._h, int(num_simulations), C.byref(fused.header), fused.jobs.data_ptr(),
                                fused.weights.data_ptr(), fused.biases.data_ptr(), fused.act_table.data_ptr(),
                                pool.data_ptr(), pool.stride(0), pool.stride(1), ix.data_ptr(), iy.data_ptr(),
                                la.data_ptr(), rew.data_ptr(), val.data_ptr(), pol.data_ptr(), int(rows_per_workgroup),
                                _stream()),
              "hz_search_run")
        self._sim += int(num_simulations)
        return ix, iy, la

    def set_predicted_lines(self, on):
        """Which persistent kernels search_tensors launches for these roots (include/hz_search.h: the ones whose descent walks
        predicted lines in trees that have grown deep -- the default -- or the plain ones).  Same results either way."""
        check(lib.hz_search_set_predicted_lines(self._h, int(bool(on))), "hz_search_set_predicted_lines")

    def root_stats_tensors(self, counts=None, values=None):
        """(visit counts [N, A] i32, root values [N] f32) in one launch; optional caller-owned outputs."""
        if counts is None:
            counts = torch.empty((self.root_num, self.action_num), dtype=torch.int32, device=self.device)
        if values is None:
            values = torch.empty(self.root_num, dtype=torch.float32, device=self.device)
        check(lib.hz_tree_get_root_stats(self._h, counts.data_ptr(), values.data_ptr(), _stream()),
              "hz_tree_get_root_stats")
        return counts, values

    def values_tensor(self):
        out = torch.empty(self.root_num, dtype=torch.float32, device=self.device)
        check(lib.hz_tree_get_values(self._h, out.data_ptr(), _stream()), "hz_tree_get_values")
        return out

    def trajectories_tensor(self, max_len=None):
        max_len = int(max_len or self.tree_nodes)
        out = torch.empty((self.root_num, max_len), dtype=torch.int32, device=self.device)
        check(lib.hz_tree_get_trajectories(self._h, out.data_ptr(), max_len, _stream()), "hz_tree_get_trajectories")
        return out

    def minmax_tensors(self):
        mn = torch.empty(self.root_num, dtype=torch.float32, device=self.device)
        mx = torch.empty(self.root_num, dtype=torch.float32, device=self.device)
        check(lib.hz_tree_get_minmax(self._h, mn.data_ptr(), mx.data_ptr(), _stream()), "hz_tree_get_minmax")
        return mn, mx

    def root_priors_tensor(self):
        out = torch.empty((self.root_num, self.action_num), dtype=torch.float32, device=self.device)
        check(lib.hz_tree_get_root_priors(self._h, out.data_ptr(), _stream()), "hz_tree_get_root_priors")
        return out

    def path_len_tensor(self):
        out = torch.empty(self.root_num, dtype=torch.int32, device=self.device)
        check(lib.hz_tree_get_path_len(self._h, out.data_ptr(), _stream()), "hz_tree_get_path_len")
        return out

    # -- the reference's list-returning read-outs -----------------------------------------------------
    def get_trajectories(self):
        t = self.trajectories_tensor().cpu().numpy()
        return [[int(a) for a in row if a >= 0] for row in t]

    def get_distributions(self):
        return self.distributions_tensor().cpu().numpy().tolist()

    def get_values(self):
        return self.values_tensor().cpu().numpy().tolist()


def multi_traverse(roots, pb_c_base, pb_c_init, discount, min_max_stats_lst, results):
    roots.set_params(pb_c_base, pb_c_init, discount, min_max_stats_lst.value_delta_max)
    results.roots = roots
    ix, iy, la = roots.traverse_tensors()
    packed = torch.stack((ix, iy, la)).cpu().numpy()
    return packed[0].tolist(), packed[1].tolist(), packed[2].tolist()


def multi_back_propagate(hidden_state_index_x, discount, rewards, values, policies, min_max_stats_lst, results):
    roots = results.roots
    if roots is None:
        raise RuntimeError("multi_back_propagate: `results` was not filled by multi_traverse")
    p = roots._params
    roots.set_params(p[0], p[1], discount, min_max_stats_lst.value_delta_max)
    roots.backprop_tensors(hidden_state_index_x, rewards, values, policies)
