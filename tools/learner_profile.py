#!/usr/bin/env python3
"""tools/learner_profile.py -- the learner step (learner.GraphedUpdate, batch 256, 5 unroll steps) in a loop, for rocprofv3:
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_learner -- python3 tools/learner_profile.py --game Hanabi-Full-5p
Prints ms per step (graph replay only, batch resident) and the number of kernels one step launches."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.learner import GraphedUpdate, make_optimizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="Hanabi-Full-5p")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--parallel-heads", type=int, default=0, help="--fused: head chains on streams of their own (FusedTrainNet)")
    ap.add_argument("--fused", action="store_true", help="the module forward through the fused Linear + BatchNorm + ReLU blocks (hanabizero_amd/fused_train.py)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = make_config(args.game, simulations=50, stack=4, p_mcts_num=256, batch_size=256)
    net = cfg.get_uniform_network().to(dev)
    if args.fused:
        from hanabizero_amd.fused_train import FusedTrainNet
        net = FusedTrainNet(net, unroll_steps=cfg.num_unroll_steps, parallel_heads=args.parallel_heads)
    opt = make_optimizer(net, cfg, capturable=True)
    g = GraphedUpdate(net, opt, cfg, cfg.batch_size)
    B, U, A, stack = cfg.batch_size, cfg.num_unroll_steps, cfg.action_space_size, cfg.stacked_observations
    D = cfg.obs_shape // stack
    rng = np.random.RandomState(0)
    pol = rng.dirichlet([0.3] * A, (B, U + 1)).astype(np.float32)
    batch = (((rng.rand(B, stack + U, D) < 0.2).astype(np.uint8), rng.randint(0, A, (B, U)), np.ones((B, U), np.float32), np.arange(B),
              np.ones(B, np.float32), np.zeros(B)), (rng.randint(0, 2, (B, U + 1)).astype(np.float32), (rng.rand(B, U + 1) * 20).astype(np.float32), pol))
    for _ in range(3):
        g(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g(batch)
    torch.cuda.synchronize()
    with_copy = (time.perf_counter() - t0) / args.steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        g._graph.replay()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"game": args.game, "blocks": "fused" if args.fused else "autograd under autocast", "ms_per_step_with_batch_copy_and_readback": 1e3 * with_copy, "ms_per_graph_replay": 1e3 * replay}))


if __name__ == "__main__":
    main()
