"""hanabizero_amd.evaluate -- the evaluation loop on the MI355X engine.

Replaces ``test()`` (/root/reference/core/test.py:41-127): the third caller of the search kernels.
Same protocol: ``test_episodes`` games seeded 0..E-1 (:50-51), every move = initial inference on the stacked
observations, ``prepare_no_noise`` (:93), ``MCTS.run_multi``, deterministic ``select_action`` (:105, first arg-max of
the legal-masked visit counts), ``env.step``; a game's final score is ``info['score']`` at its terminal step (:117-118).
The reference hard-codes 1000 episodes (:42); here it is an argument defaulting to that.

r04: the loop IS the self-play actor's lock-step (hanabizero_amd.selfplay.SelfPlayActor with root_noise=False,
deterministic=True) -- root inference, k_prepare, ONE persistent search kernel, the two move-tail launches, replayed as a
hipGraph -- instead of an eager loop with two host syncs per move.  The reference keeps every env in the batch until the last
game has ended and merely stops stepping the finished ones (:99-100; its model and MCTS still run on all of them); here a
finished env starts another game, which costs the same and is ignored: an episode's result is the FIRST finished game of its
env (the records carry the env id).  Completion is looked at every `check_every` moves, when the finished games' records are
drained anyway.
"""
import numpy as np
import torch

from .selfplay import SelfPlayActor


last_run = {}  # lock-steps / envs of the last test() call (tools/next_rows_bench.py reports the batch's rate beside the episodes')


def test(config, engine, counter=0, test_episodes=1000, device=None, tie_seed=0, max_moves=None, use_graph=True, check_every=8):
    """Returns (ep_final_rewards list[int], ep_steps list[int])."""
    import time
    t_start = time.perf_counter()
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    E = int(test_episodes)
    from ._lib import poll_giveups
    fused = getattr(engine, "fused", None) is not None
    giveups_before = poll_giveups() if fused else 0
    # (a game lasts at least one move: at most `check_every` games per env between two drains)
    actor = SelfPlayActor(config, engine, E, seed=0, device=device, use_graph=use_graph, deterministic=True, root_noise=False,
                          tie_seed=tie_seed, outbox_games=max((check_every + 3) * E, 1024))
    limit = max_moves if max_moves is not None else 10 * config.max_moves
    final = np.full(E, -1, np.int64)
    steps = np.zeros(E, np.int64)
    with torch.no_grad():
        if use_graph:           # (the capture plays its three moves now: `last_run` times it apart from the replays)
            actor.step()
            torch.cuda.synchronize(device)
        t_loop, moves_at_loop = time.perf_counter(), actor.total_moves
        while (final < 0).any() and actor.total_moves // E < limit:
            before = actor.total_moves
            while actor.total_moves - before < check_every * E:  # (the first step() of a graph actor plays three moves)
                actor.step()
            rec = actor.drain()
            if rec is None:
                continue
            meta = rec["meta"]  # [games, 4]: length, final score, env id, -- in the order the games ended
            env, first = np.unique(meta[:, 2], return_index=True)
            new = final[env] < 0
            final[env[new]], steps[env[new]] = meta[first[new], 1], meta[first[new], 0]
    assert int(actor.illegal_steps) == 0, "evaluation produced an illegal move"
    if fused and poll_giveups() != giveups_before:  # (include/hz_mlp.h: must not happen)
        raise RuntimeError("the fused inference gave up waits on its arrival counters during this evaluation: its searches "
                           "ran with inputs that may not have been there")
    torch.cuda.synchronize(device)
    last_run.update(lock_steps=actor.total_moves // E, envs=E, setup_and_capture_s=t_loop - t_start, replay_s=time.perf_counter() - t_loop,
                    replayed_lock_steps=(actor.total_moves - moves_at_loop) // E)
    done = final >= 0  # (episodes still running at `max_moves` report score 0 after that many steps)
    return np.where(done, final, 0).tolist(), np.where(done, steps, actor.total_moves // E).tolist()
