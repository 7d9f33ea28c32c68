"""Where a drain of finished games spends its time: the count read, the pack kernel, the device-to-host copy.
usage: python tools/drain_profile.py [workload=full4096] [steps between drains=40]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.dist import gather_packed, reserve_landing  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "full4096"
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    game, N, S, stack = bench.WORKLOADS[workload]
    device = torch.device("cuda", 0)
    cfg = make_config(game, simulations=S, stack=stack, p_mcts_num=N)
    engine = bench.build_engine(cfg, torch.bfloat16, device, fused=None)
    actor = SelfPlayActor(cfg, engine, N, seed=0, device=device, use_graph=True)
    reserve_landing(16384 * N)
    for rnd in range(3):
        for _ in range(every):
            actor.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        packed = actor.drain_packed()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        got = gather_packed(packed, actor.A, actor.W)
        t3 = time.perf_counter()
        nbytes = packed[0].numel()
        print("drain %d: %d games, %d moves, %.1f MB: drain_packed host %.3f ms (+%.3f ms until the pack kernel is done), "
              "copy to pinned host %.3f ms = %.1f GB/s" % (rnd, packed[1], packed[2], nbytes / 1e6, 1e3 * (t1 - t0), 1e3 * (t2 - t1),
                                                            1e3 * (t3 - t2), nbytes / (t3 - t2) / 1e9), flush=True)
        del got


if __name__ == "__main__":
    main()
