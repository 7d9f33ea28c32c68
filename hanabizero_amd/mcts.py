"""hanabizero_amd.mcts -- ``MCTS(config).run_multi(roots, model, hidden_state_roots)`` of
/root/reference/core/mcts.py:7-57, device-resident.

The reference round-trips every simulation through the host (per-tree Python gather mcts.py:31-32, 2 H2D and 4 D2H
copies, .tolist() conversions mcts.py:45-50).  Here the hidden-state pool [num_simulations, N, H] stays in HBM, the
traverse kernel gathers each tree's parent state straight into the dynamics net's input buffer, the nets are
invoked once per simulation over all N trees, and their outputs feed the backup kernel without leaving the device.
Same quirks as the reference: num_simulations - 1 real simulations (mcts.py:24-26), NaN logits -> 0 (mcts.py:48-49).
"""
import numpy as np
import torch

from . import cytree as tree
from .model import InferenceEngine


class MCTS(object):
    def __init__(self, config, persistent=True, rows_per_workgroup=0):
        self.config = config
        self.persistent = persistent  # fused engines: the whole simulation loop as one persistent kernel
        self.rows_per_workgroup = rows_per_workgroup  # of that kernel: 0 = auto, 16 / 32 / -32 force a shape (tests, tools)

    def run_multi(self, roots, model, hidden_state_roots, pool=None):
        """roots: hanabizero_amd.cytree.Roots (already prepared).  model: an InferenceEngine (fast path) or a
        module with the reference's recurrent_inference (numpy outputs; compatibility path).
        hidden_state_roots: [N, H] CUDA tensor or numpy array.  Mutates `roots`; returns None."""
        cfg = self.config
        with torch.no_grad():
            num, S = roots.num, cfg.num_simulations
            roots.set_params(cfg.pb_c_base, cfg.pb_c_init, cfg.discount, cfg.value_delta_max)
            if not isinstance(model, InferenceEngine):
                return self._run_multi_compat(roots, model, hidden_state_roots)
            h0 = hidden_state_roots if isinstance(hidden_state_roots, torch.Tensor) else torch.as_tensor(
                np.asarray(hidden_state_roots), device=roots.device)
            H = h0.shape[1]
            if pool is None:
                pool = torch.empty((S, num, H), dtype=model.dtype, device=roots.device)
            if h0.data_ptr() != pool[0].data_ptr():  # (the actor has the root inference write plane 0 directly)
                pool[0].copy_(h0)
            oh = model.onehot_cols
            net_in = torch.empty((num, H + oh), dtype=model.dtype, device=roots.device)
            fused = getattr(model, "fused", None)
            if fused is not None:
                # traverse+gather (HIP) -> whole recurrent inference as one MFMA kernel (HIP) -> expand/backup (HIP)
                rew = torch.empty(num, dtype=torch.float32, device=roots.device)
                val = torch.empty(num, dtype=torch.float32, device=roots.device)
                pol = torch.empty((num, roots.action_num), dtype=torch.float32, device=roots.device)
                fused16 = model.fused_shape(16, 2) if self.persistent else None
                # (the persistent kernels are written for < 64 simulations -- the reference's configs have 50 -- with no loops over
                # more in their simulation loop: include/hz_search.h; beyond that the launch-per-phase search below computes the
                # same bits)
                # (hz_search_run's own limits, mirrored: tree capacity < 64 simulations, <= 16 passes, hidden <= 512, 16 rows of
                # the image + the 16 trees' search state within 160 KiB -- anything else takes the launch-per-phase search below)
                if fused16 is not None and S < 64 and fused16.n_jobs <= 16 and H <= 512 and \
                        fused16.lds_bytes(16) + 16 * ((S + 1) * 20 + S * 4) + (16 + 2) * 4 + 32 * 8 + 128 * 4 + 128 <= 160 * 1024:
                    # the whole loop below as ONE persistent kernel: a workgroup keeps 16 trees for all simulations
                    roots.search_tensors(fused16, pool, S - 1, rew, val, pol, self.rows_per_workgroup)
                    return
                # 2 launches per simulation: [MFMA recurrent inference] [backup of sim k + descent of sim k+1]
                ix, _, la = roots.traverse_tensors()  # the MFMA kernel gathers pool[ix, tree] itself
                for index_simulation in range(S - 1):
                    fused(pool, ix, la, pool[index_simulation + 1], rew, val, pol)
                    if index_simulation < S - 2:
                        ix, _, la = roots.backprop_traverse_tensors(index_simulation + 1, rew, val, pol)
                    else:
                        roots.backprop_tensors(index_simulation + 1, rew, val, pol)
                return
            for index_simulation in range(S - 1):
                # select + gather + one-hot (1 kernel) -> dynamics/prediction GEMMs -> scalar transform + NaN
                # clearing + expand + backup + min-max (1 kernel)
                roots.traverse_tensors(pool, net_in, onehot_cols=oh)
                r_log, v_log, p_log = model.recurrent_heads(net_in, pool[index_simulation + 1])
                roots.backprop_nets_tensors(index_simulation + 1, r_log, v_log, model.V, -model.support, p_log)

    def _run_multi_compat(self, roots, model, hidden_state_roots):
        """The reference loop verbatim in structure (lists / numpy through the drop-in cytree API): lets an unmodified
        reference model object drive the HIP tree."""
        cfg = self.config
        model.eval()
        num = roots.num
        hidden_state_pool = [np.asarray(hidden_state_roots)]
        hidden_state_index_x = 0
        mm = tree.MinMaxStatsList(num)
        mm.set_delta(cfg.value_delta_max)
        dev = next(model.parameters()).device
        for index_simulation in range(cfg.num_simulations):
            if index_simulation == cfg.num_simulations - 1:
                continue
            results = tree.ResultsWrapper(num)
            ix_lst, iy_lst, last_actions = tree.multi_traverse(roots, cfg.pb_c_base, cfg.pb_c_init, cfg.discount, mm, results)
            hidden_states = np.asarray([hidden_state_pool[ix][iy] for ix, iy in zip(ix_lst, iy_lst)])
            hidden_states = torch.from_numpy(hidden_states).to(dev)
            la = torch.from_numpy(np.asarray(last_actions)).to(dev).unsqueeze(1).long()
            out = model.recurrent_inference(hidden_states, la)
            logits = np.array(out.policy_logits)
            logits[np.isnan(logits)] = 0.0
            hidden_state_pool.append(out.hidden_state)
            hidden_state_index_x += 1
            tree.multi_back_propagate(hidden_state_index_x, cfg.discount, np.asarray(out.reward).reshape(-1).tolist(),
                                      np.asarray(out.value).reshape(-1).tolist(), logits.tolist(), mm, results)
