#!/usr/bin/env python3
"""tools/loop_bench.py -- BASELINE.json configs[4] timed as ONE loop on ONE GPU, one JSON line.

Hanabi-Full 5 players (A = 48, D = 1385, mdp global), 50 simulations per move: self-play (SelfPlayActor under its hipGraph) ->
drain -> ReplayBuffer.ingest_packed -> prioritised sampling -> reanalyze of a share of every batch with the target model
(policy_re_context + prepare_policy_re: the search kernels' second caller) -> make_batch -> GraphedUpdate (batch 256, 5 unroll
steps, bf16 autocast) -> priorities back -> every checkpoint_interval learner steps the actor's engine takes the learner's
weights in place.  What the reference runs as Ray actors (/root/reference/core/train.py:317-431, reanalyze_worker.py:307-422,
selfplay_worker.py:91-393) is here a single synchronous loop: no control plane, no service.

The schedule is the reference's replay ratio (README.md:51: 0.008 learner steps per self-play move): after every lock-step the
loop owes `envs * ratio` learner steps and pays them before the next one.  Reported: self-play moves/s and learner steps/s of the
whole loop, the ratio achieved, and where the wall time went (self-play launch + wait, drain + ingest, sampling, reanalyze,
make_batch, update) -- which answers whether the search is what limits this configuration (it is not: the learner side is).
--ratio 0 runs self-play + ingest only (what the actors alone sustain)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.dist import gather_packed  # noqa: E402
from hanabizero_amd.learner import GraphedUpdate, adjust_lr, make_batch, make_optimizer  # noqa: E402
from hanabizero_amd.model import InferenceEngine  # noqa: E402
from hanabizero_amd.reanalyze import policy_re_context, prepare_policy_re  # noqa: E402
from hanabizero_amd.replay import ReplayBuffer  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="Hanabi-Full-5p")
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--lock-steps", type=int, default=40)
    ap.add_argument("--warm-steps", type=int, default=60, help="self-play only, to fill the replay buffer")
    ap.add_argument("--ratio", type=float, default=0.008, help="learner steps per self-play move (reference README.md:51)")
    ap.add_argument("--reanalyze-share", type=float, default=0.5, help="share of every batch whose policy targets are re-searched")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--flush-every", type=int, default=10)
    ap.add_argument("--obs-float32", action="store_true", help="make_batch in the reference's float32 layout (default: frames stay bytes until they are on the device)")
    ap.add_argument("--checkpoint-interval", type=int, default=0, help="learner steps between weight hand-overs to the actor (0: the config's, 2000 for Hanabi-Full; one hand-over is timed after the loop either way)")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    cfg = make_config(args.game, simulations=50, stack=4, p_mcts_num=args.envs, batch_size=256)
    engine = bench.build_engine(cfg, dtype, device)
    target = bench.build_engine(cfg, dtype, device)  # the reanalyze workers' target model
    actor = SelfPlayActor(cfg, engine, args.envs, seed=0, device=device)
    rb = ReplayBuffer(cfg)
    t = dict(selfplay=0.0, drain=0.0, sample=0.0, reanalyze=0.0, make_batch=0.0, update=0.0, weights=0.0)

    def drain():
        t0 = time.perf_counter()
        for buf, n, moves in gather_packed(actor.drain_packed(), actor.A, actor.W):
            rb.ingest_packed(buf, n, moves)
        t["drain"] += time.perf_counter() - t0

    for k in range(args.warm_steps):
        actor.step()
        if (k + 1) % args.flush_every == 0:
            drain()
    torch.cuda.synchronize()
    drain()
    learner = cfg.get_uniform_network().to(device)
    learner.load_state_dict(engine._net.state_dict())
    handover = cfg.get_uniform_network()  # host copy the weights travel through (selfplay_worker.py:177-184: set_weights)
    handover.eval()
    opt = make_optimizer(learner, cfg, capturable=True)
    graphed = GraphedUpdate(learner, opt, cfg, cfg.batch_size)
    value_fn = lambda o: target.initial(torch.from_numpy(o).to(device))[0].float().cpu().numpy()
    R = int(cfg.batch_size * args.reanalyze_share)

    def learner_step(it):
        t0 = time.perf_counter()
        games, pos, idx, w, mt = rb.prepare_batch_context(cfg.batch_size, beta=0.4)
        t1 = time.perf_counter()
        pol_re = None
        if R:
            ctx = policy_re_context(cfg, games[:R], pos[:R], idx[:R])
            pol_re = prepare_policy_re(cfg, target, ctx, tie_seed=it)
        t2 = time.perf_counter()
        batch = make_batch(games, pos, cfg, value_fn, weights=w, rng=np.random.RandomState(it), policy_re=pol_re, obs_dtype=np.float32 if args.obs_float32 else np.uint8)
        t3 = time.perf_counter()
        adjust_lr(cfg, opt, it)
        loss_data, prio = graphed(batch)
        rb.update_priorities(idx, prio, mt)
        t4 = time.perf_counter()
        t["sample"] += t1 - t0
        t["reanalyze"] += t2 - t1
        t["make_batch"] += t3 - t2
        t["update"] += t4 - t3
        return loss_data

    for it in range(2):  # (captures the learner's graph, first shapes of the reanalyze search)
        learner_step(it)
    torch.cuda.synchronize()
    for k in t:
        t[k] = 0.0
    owed, steps_done, losses = 0.0, 0, []
    games_before = int(actor.out_count[0].item())
    t_start = time.perf_counter()
    for k in range(args.lock_steps):
        t0 = time.perf_counter()
        actor.step()
        if args.ratio > 0:
            torch.cuda.synchronize()  # (a synchronous loop: the learner's work below does not overlap the move)
        t["selfplay"] += time.perf_counter() - t0
        if (k + 1) % args.flush_every == 0:
            drain()
        owed += args.envs * args.ratio
        while owed >= 1.0:
            losses.append(learner_step(2 + steps_done)[1])
            steps_done += 1
            owed -= 1.0
            if steps_done % (args.checkpoint_interval or cfg.checkpoint_interval) == 0:
                t0 = time.perf_counter()
                handover.load_state_dict({k: v.detach().cpu() for k, v in learner.state_dict().items()})
                engine.load(handover)  # in place: the actor's captured graph sees the new weights from its next replay on
                t["weights"] += time.perf_counter() - t0
    torch.cuda.synchronize()
    drain()
    wall = time.perf_counter() - t_start
    t0 = time.perf_counter()  # one weight hand-over, timed on its own (outside `wall` unless the interval fell inside the run)
    handover.load_state_dict({k: v.detach().cpu() for k, v in learner.state_dict().items()})
    engine.load(handover)
    torch.cuda.synchronize()
    handover_ms = 1e3 * (time.perf_counter() - t0)
    for _ in range(3):
        actor.step()
    torch.cuda.synchronize()
    moves = args.envs * args.lock_steps
    out = {"workload": "%s, %d envs, 50 sims/move, %s nets: self-play + reanalyze (%.0f %% of each batch) + learner batch %d on ONE GPU, one synchronous loop"
                       % (args.game, args.envs, args.dtype, 100 * args.reanalyze_share, cfg.batch_size),
           "lock_steps": args.lock_steps, "selfplay_moves_per_s": moves / wall, "learner_steps_per_s": steps_done / wall,
           "learner_steps": steps_done, "replay_ratio_target": args.ratio, "replay_ratio_achieved": steps_done / moves,
           "reference": {"replay_ratio": 0.008, "learner_steps_per_s": 1000 / 160.0, "selfplay_moves_per_s_derived": 1000 / 160.0 / 0.008,
                         "hardware": "4 x RTX 3090 + 96 CPU cores, Hanabi-Small", "source": "/root/reference/README.md:51"},
           "wall_s": wall, "games_finished": int(actor.out_count[0].item()) - games_before, "replay_positions": rb.get_total_len(),
           "ms_per_lock_step": {k: 1e3 * v / args.lock_steps for k, v in t.items()},
           "ms_per_learner_step": ({k: 1e3 * t[k] / steps_done for k in ("sample", "reanalyze", "make_batch", "update")} if steps_done else None),
           "weight_handover_ms": handover_ms, "checkpoint_interval": args.checkpoint_interval or cfg.checkpoint_interval,
           "loss_first_last": [float(losses[0]), float(losses[-1])] if losses else None,
           "illegal_steps": int(actor.illegal_steps)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
