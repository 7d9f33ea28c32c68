"""Lock-step time with one vs two trees per wavefront of the persistent search kernel (hz_search_run's rows_per_workgroup argument).
usage: python tools/search_rows.py [workload=full8192] [steps=12]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from hanabizero_amd._lib import check, lib  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "full8192"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    game, N, S, stack = bench.WORKLOADS[workload]
    device = torch.device("cuda", 0)
    cfg = make_config(game, simulations=S, stack=stack, p_mcts_num=N)
    engine = bench.build_engine(cfg, torch.bfloat16, device, fused=None)
    for rows in (16, -32, 32):  # one tree per wave; two, one after the other; two, side by side in the halves
        actor = SelfPlayActor(cfg, engine, N, seed=0, device=device, use_graph=True)
        actor.mcts.rows_per_workgroup = rows
        actor._capture()
        for _ in range(3):
            actor.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            actor.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("%s rows/workgroup %d: %.3f ms/step = %.0f moves/s" % (workload, rows, 1e3 * dt, N / dt), flush=True)
        del actor


if __name__ == "__main__":
    main()
