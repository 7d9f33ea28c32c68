#!/usr/bin/env python3
"""tools/next_rows_bench.py -- timings of the callers either side of the hot path on one MI355X,
Hanabi-Full 2p, bf16 nets, random-init weights, synthetic data of the reference's shapes:
  f-1 reanalyze policy targets (reanalyze.prepare_policy_re; reference batch: 256 positions x (5 unroll steps + 1) = 1536 roots)
  f-2 replay ingest            (ReplayBuffer.ingest_packed of the actor's packed records)
  f-3 learner step             (learner.update_weights, batch 256, bf16 autocast; make_batch on the host beside it)
  f-4 evaluation               (evaluate.test, 1000 episodes)
Prints one JSON object.  usage: python tools/next_rows_bench.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.dist import gather_packed  # noqa: E402
from hanabizero_amd.evaluate import test as run_test  # noqa: E402
from hanabizero_amd.learner import GraphedUpdate, make_batch, make_optimizer, update_weights  # noqa: E402
from hanabizero_amd.reanalyze import prepare_policy_re  # noqa: E402
from hanabizero_amd.replay import ReplayBuffer  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    device = torch.device("cuda", 0)
    cfg = make_config("Hanabi-Full", simulations=50, stack=4, p_mcts_num=4096)
    cfg.batch_size = 256
    engine = bench.build_engine(cfg, torch.bfloat16, device, fused=None)
    A, U = cfg.action_space_size, cfg.num_unroll_steps + 1
    out = {}

    # f-1: reanalyze policy targets
    rng = np.random.RandomState(0)
    for P in (256, 2048):
        B = P * U
        obs = torch.from_numpy((rng.rand(B, cfg.obs_shape) < 0.15).astype(np.float32)).to(device)
        legal = (rng.rand(B, A) < 0.6).astype(np.float64)
        legal[:, 0] = 1
        ctx = (obs, np.ones(B, np.int64), list(range(P)), list(range(P)), None, None, legal)
        dt = timed(lambda: prepare_policy_re(cfg, engine, ctx, tie_seed=1), 5)
        out["reanalyze_%d_positions" % P] = {"roots": B, "ms": 1e3 * dt, "roots_per_s": B / dt, "sims_per_s": B * (cfg.num_simulations - 1) / dt}

    # self-play data for f-2 / f-3
    actor = SelfPlayActor(cfg, engine, 4096, seed=0, device=device)
    for _ in range(40):
        actor.step()
    torch.cuda.synchronize()
    got = gather_packed(actor.drain_packed(), actor.A, actor.W)
    rb = ReplayBuffer(cfg)
    t0 = time.perf_counter()
    games = sum(rb.ingest_packed(buf, n, moves) for buf, n, moves in got)
    dt = time.perf_counter() - t0
    out["replay_ingest"] = {"games": games, "positions": rb.get_total_len(), "ms": 1e3 * dt, "games_per_s": games / dt,
                            "note": "host Python: unpack + GameHistory objects + priorities"}

    # f-3: learner
    learner = cfg.get_uniform_network().to(device)
    opt = make_optimizer(learner, cfg)
    value_fn = lambda o: engine.initial(torch.from_numpy(o).to(device))[0].float().cpu().numpy()
    g, pos, idx, w, mt = rb.prepare_batch_context(cfg.batch_size, beta=0.4)
    batch = make_batch(g, pos, cfg, value_fn, weights=w, rng=np.random.RandomState(0))  # (warm-up: first inference of this shape)
    t0 = time.perf_counter()
    for rep in range(5):
        g, pos, idx, w, mt = rb.prepare_batch_context(cfg.batch_size, beta=0.4)
        batch = make_batch(g, pos, cfg, value_fn, weights=w, rng=np.random.RandomState(rep))
    out["make_batch_256"] = {"ms": 1e3 * (time.perf_counter() - t0) / 5,
                             "note": "prioritised sampling + host assembly + one target-model inference, mean of 5"}
    dt = timed(lambda: update_weights(learner, batch, opt, cfg, amp=torch.bfloat16), 10)
    out["update_weights_256"] = {"ms": 1e3 * dt, "steps_per_s": 1 / dt, "samples_per_s": cfg.batch_size / dt}
    learner2 = cfg.get_uniform_network().to(device)
    graphed = GraphedUpdate(learner2, make_optimizer(learner2, cfg, capturable=True), cfg, cfg.batch_size)
    dt = timed(lambda: graphed(batch), 20)
    out["graphed_update_256"] = {"ms": 1e3 * dt, "steps_per_s": 1 / dt, "samples_per_s": cfg.batch_size / dt,
                                 "note": "the same step as one hipGraph replay (learner.GraphedUpdate), incl. the H2D copy of the batch"}
    del actor

    # f-4: evaluation
    t0 = time.perf_counter()
    scores, steps = run_test(cfg, engine, test_episodes=1000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from hanabizero_amd import evaluate
    out["evaluation_1000_episodes"] = {"s": dt, "moves": int(np.sum(steps)), "moves_per_s": float(np.sum(steps)) / dt,
                                       "mean_score": float(np.mean(scores)), "longest_game": int(np.max(steps)),
                                       # the batch keeps all 1000 envs moving until the longest episode has ended (as the reference's loop
                                       # keeps them in its model / MCTS batch, core/test.py:99-100): the lock-steps' own rate, incl. capture
                                       "lock_steps": evaluate.last_run.get("lock_steps"),
                                       "batch_moves_per_s": evaluate.last_run.get("lock_steps", 0) * 1000 / dt,
                                       # ... and of the replayed lock-steps alone (random-init games last 8 moves: set-up and capture are most of `s`)
                                       "setup_and_capture_s": evaluate.last_run.get("setup_and_capture_s"),
                                       "replayed_batch_moves_per_s": evaluate.last_run.get("replayed_lock_steps", 0) * 1000 / max(1e-9, evaluate.last_run.get("replay_s", 0))}
    t0 = time.perf_counter()
    scores4, steps4 = run_test(cfg, engine, test_episodes=4096)
    dt = time.perf_counter() - t0
    out["evaluation_4096_episodes"] = {"s": dt, "moves": int(np.sum(steps4)), "moves_per_s": float(np.sum(steps4)) / dt,
                                       "mean_score": float(np.mean(scores4)), "lock_steps": evaluate.last_run.get("lock_steps"),
                                       "batch_moves_per_s": evaluate.last_run.get("lock_steps", 0) * 4096 / dt,
                                       "setup_and_capture_s": evaluate.last_run.get("setup_and_capture_s"),
                                       "replayed_batch_moves_per_s": evaluate.last_run.get("replayed_lock_steps", 0) * 4096 / max(1e-9, evaluate.last_run.get("replay_s", 0))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
