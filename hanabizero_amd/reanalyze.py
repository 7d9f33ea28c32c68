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


_giveups_seen = None  # hz_mlp_poll_giveups at the last check (per process; read before the first search)


def policy_re_context(config, games, positions, indices=None):
    """What ``BatchWorker_CPU._prepare_policy_re_context`` (reanalyze_worker.py:101-144) hands to the searching worker for the
    sampled (game, position) pairs whose policy targets are to be refreshed: for each of the num_unroll_steps + 1 unroll
    positions the stacked observation window and the legal-action mask, or a zero window / all-illegal mask and
    policy_mask 0 past the end of the game.  Returns the reference's 7-tuple with policy_obs_lst as one array
    [B', stack, D] (the reference wraps it in a Ray ObjectRef)."""
    U, stack, A = config.num_unroll_steps, config.stacked_observations, config.action_space_size
    D = config.obs_shape // stack
    n = len(games)
    obs = np.zeros((n * (U + 1), stack, D), np.float32)
    legal = np.zeros((n * (U + 1), A), np.float64)
    mask = np.zeros(n * (U + 1), np.int64)
    traj_lens, child_visits = [], []
    k = 0
    for game, pos in zip(games, positions):
        T = len(game)
        traj_lens.append(T)
        child_visits.append(game.child_visits)
        frames = np.asarray(game.obs(pos, U))
        for cur in range(pos, pos + U + 1):
            if cur < T:
                mask[k] = 1
                obs[k] = frames[cur - pos:cur - pos + stack]
                legal[k] = game.legal_actions[cur]
            k += 1
    return obs, mask.tolist(), list(positions), list(range(n)) if indices is None else list(indices), child_visits, traj_lens, legal


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
    global _giveups_seen
    if _giveups_seen is None and getattr(engine, "fused", None) is not None:
        from ._lib import poll_giveups
        _giveups_seen = poll_giveups()
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
        if getattr(engine, "fused", None) is not None:  # (the read-back above has synchronised: this costs a 4-byte copy)
            from ._lib import poll_giveups
            now = poll_giveups()
            if now != _giveups_seen:
                grew, _giveups_seen = now - _giveups_seen, now
                raise RuntimeError("the fused inference gave up %d wait(s) on its arrival counters (include/hz_mlp.h): these policy "
                                   "targets were searched with inputs that may not have been there" % grew)
    U = config.num_unroll_steps + 1
    assert B == len(state_index_lst) * U, "policy_re_context holds (num_unroll_steps + 1) entries per sampled position"
    return policy.reshape(len(state_index_lst), U, A)
