#!/usr/bin/env python3
"""tools/timing_outliers.py [workload] [reps] -- bench.py's timing pass of the search kernel over and over: every graph's time,
and the count of arrival-counter waits that gave up (hz_mlp_poll_giveups).  For chasing a launch that took 30 ms once."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from hanabizero_amd._lib import poll_giveups
    from hanabizero_amd.config import make_config
    from hanabizero_amd.selfplay import SelfPlayActor
    wl = sys.argv[1] if len(sys.argv) > 1 else "full8192"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    game, N, sims, stack = bench.WORKLOADS[wl]
    cfg = make_config(game, simulations=sims, stack=stack, p_mcts_num=N)
    eng = bench.build_engine(cfg, torch.bfloat16, "cuda")
    actor = SelfPlayActor(cfg, eng, num_envs=N, rank=0, seed=1, use_graph=True)
    for _ in range(3):
        actor.step()
    torch.cuda.synchronize()
    worst = 0.0
    for r in range(reps):
        s = bench.kernel_timing(actor, sample_sims=())[1]
        print("rep %2d: mean %.1f us  best %.1f us  giveups %d" % (r, s["mean_s"] * 1e6, s["min_s"] * 1e6, poll_giveups()), flush=True)
        worst = max(worst, s["mean_s"])
        for _ in range(5):
            actor.step()
    print("worst mean %.1f us, giveups %d" % (worst * 1e6, poll_giveups()))


if __name__ == "__main__":
    main()
