"""hanabizero_amd.evaluate -- the evaluation loop on the MI355X engine.

Replaces ``test()`` (/root/reference/core/test.py:41-127): the third caller of the search kernels.
Same protocol: ``test_episodes`` games seeded 0..E-1 (:50-51), every move = initial inference on the stacked
observations, ``prepare_no_noise`` (:93), ``MCTS.run_multi``, deterministic ``select_action`` (:105, first arg-max of
the legal-masked visit counts), ``env.step``; finished games are no longer stepped (:99-100) and their final score is
``info['score']`` at the terminal step (:117-118).  The reference hard-codes 1000 episodes (:42); here it is an argument
defaulting to that.  Everything stays on the device; one small D2H copy per move checks for completion.
"""
import ctypes as C

import numpy as np
import torch

from . import cytree
from ._lib import check, lib
from .hanabi_env import HanabiVecEnv
from .mcts import MCTS


def test(config, engine, counter=0, test_episodes=1000, device=None, tie_seed=0, max_moves=None):
    """Returns (ep_final_rewards list[int], ep_steps list[int])."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    E, A, S, stack = int(test_episodes), config.action_space_size, config.num_simulations, config.stacked_observations
    env = HanabiVecEnv(config.env_name, np.arange(E), device=device, mdp=config.mdp)
    D = env.obs_dim
    stack_buf = torch.zeros((E, stack, D), dtype=engine.dtype, device=device)
    newest = torch.zeros((E, D), dtype=engine.dtype, device=device)
    legal = torch.zeros((E, A), dtype=torch.uint8, device=device)
    env.reset()
    env.observe(out=newest, legal=legal)
    stack_buf.copy_(newest[:, None, :].expand(-1, stack, -1))  # GameHistory.init with the reset obs repeated (:63-64)
    done = torch.zeros(E, dtype=torch.bool, device=device)
    final = torch.zeros(E, dtype=torch.int32, device=device)
    steps = torch.zeros(E, dtype=torch.int32, device=device)
    action = torch.zeros(E, dtype=torch.int32, device=device)
    roots = cytree.Roots(E, A, S, device=device, tie_seed=tie_seed)
    mcts = MCTS(config)
    zeros = torch.zeros(E, dtype=torch.float32, device=device)
    stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    limit = max_moves if max_moves is not None else 10 * config.max_moves
    from ._lib import poll_giveups
    giveups_before = poll_giveups() if getattr(engine, "fused", None) is not None else 0
    with torch.no_grad():
        for _ in range(limit):
            if bool(done.all()):
                break
            _, logits, hidden = engine.initial(stack_buf.view(E, stack * D))
            roots.prepare_no_noise(zeros, logits, legal)
            mcts.run_multi(roots, engine, hidden)
            counts = roots.distributions_tensor()
            check(lib.hz_select_action(E, A, counts.data_ptr(), legal.data_ptr(), None, 1.0, 1, action.data_ptr(), None,
                                       stream()), "hz_select_action")
            active = ~done
            reward, d, score, status = env.step(action, active)
            assert int(((status != 0) & active).sum()) == 0, "evaluation produced an illegal move"
            steps += active.to(torch.int32)
            newly = active & d.bool()
            final = torch.where(newly, score, final)
            done = done | newly
            env.observe(out=newest, legal=legal)  # finished games keep their last observation; they are not stepped again
            shifted = torch.cat((stack_buf[:, 1:], newest[:, None, :]), dim=1)
            stack_buf.copy_(torch.where(active[:, None, None], shifted, stack_buf))
    scores, lengths = final.cpu().numpy().tolist(), steps.cpu().numpy().tolist()
    if getattr(engine, "fused", None) is not None and poll_giveups() != giveups_before:  # (include/hz_mlp.h: must not happen)
        raise RuntimeError("the fused inference gave up waits on its arrival counters during this evaluation: its searches "
                           "ran with inputs that may not have been there")
    return scores, lengths
