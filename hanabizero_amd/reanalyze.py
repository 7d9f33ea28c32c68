"""hanabizero_amd.reanalyze -- policy-target refresh of the reanalyze worker on the MI355X engine.

Replaces ``BatchWorker_GPU._prepare_policy_re`` (/root/reference/core/reanalyze_worker.py:307-371): the second caller of
the search kernels.  Same inputs and output as the reference; what happens in between stays
on the device: initial inference on all B*(num_unroll_steps+1) stacked observations, ``Roots.prepare`` with the
Dirichlet noise pre-masked by the legal actions (:344) and the initial inference's zero reward list as ``reward_pool``
(:339-340, core/model.py:71), ``MCTS.run_multi`` with the target model, visit distributions normalised over ALL
children (the reference does not mask illegal ones here, :360-362) or zeroed where ``policy_mask`` is 0 (:357-358).
"""
import numpy as np
import torch

from . import cytree
from .mcts import MCTS


def prepare_policy_re(config, engine, policy_re_context, noises=None, generator=None, tie_seed=0, device=None):
    """policy_re_context = (policy_obs_lst, policy_mask, state_index_lst, indices, child_visits, traj_lens,
    legal_action_lst) exactly as BatchWorker_CPU builds it (reanalyze_worker.py:101-167); policy_obs_lst is an array
    [B', stacked_observations * D] (or [B', stack, D]) instead of a Ray ObjectRef.
    noises: optional [B', A] float32 Dirichlet samples (for reproducible tests); drawn on the device otherwise.
    Returns np.ndarray [len(state_index_lst), num_unroll_steps + 1, A] (reanalyze_worker.py:369-370)."""
    if policy_re_context is None:
        return []
    policy_obs_lst, policy_mask, state_index_lst, indices, child_visits, traj_lens, legal_action_lst = policy_re_context
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    obs = policy_obs_lst if isinstance(policy_obs_lst, torch.Tensor) else torch.as_tensor(np.asarray(policy_obs_lst))
    obs = obs.to(device).reshape(obs.shape[0], -1)
    B, A = obs.shape[0], config.action_space_size
    legal = torch.as_tensor(np.asarray(legal_action_lst), device=device).reshape(B, A)
    with torch.no_grad():
        _, logits, hidden = engine.initial(obs)
        if noises is None:
            alpha = torch.full((B, A), float(config.root_dirichlet_alpha), dtype=torch.float64, device=device)
            g = torch._standard_gamma(alpha, generator=generator)
            noises = (g / g.sum(1, keepdim=True)).to(torch.float32)
        else:
            noises = torch.as_tensor(np.asarray(noises), dtype=torch.float32, device=device)
        noises = noises * legal.to(torch.float32)                       # reanalyze_worker.py:344
        roots = cytree.Roots(B, A, config.num_simulations, device=device, tie_seed=tie_seed)
        roots.prepare(config.root_exploration_fraction, noises, torch.zeros(B, device=device), logits,
                      (legal != 0).to(torch.uint8))                     # mock_legal_actions :345
        MCTS(config).run_multi(roots, engine, hidden)
        dist = roots.distributions_tensor().to(torch.float64)
        policy = dist / dist.sum(1, keepdim=True)
        mask = torch.as_tensor(np.asarray(policy_mask), device=device).reshape(B, 1)
        policy = torch.where(mask != 0, policy, torch.zeros_like(policy)).cpu().numpy()
    U = config.num_unroll_steps + 1
    assert B == len(state_index_lst) * U, "policy_re_context holds (num_unroll_steps + 1) entries per sampled position"
    return policy.reshape(len(state_index_lst), U, A)
