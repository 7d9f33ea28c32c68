#!/usr/bin/env python3
"""tools/tail_profile.py -- diagnostic build of the fused move tail (-DHZ_TAIL_PROFILE): s_memrealtime stamps (100 MHz) of the
owner and partner waves of every 64th workgroup of k_move_tail_a / k_move_tail_b, one lock-step of 4096 Hanabi-Full envs.
Builds a scratch copy of the library (HANABIZERO_HIP_LIB points the loader at it).  Never quote run times of it."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _build import build  # noqa: E402
out = build("libtail_prof.so", {"hz_movetail.hip": ["-DHZ_TAIL_PROFILE"]})
os.environ["HANABIZERO_HIP_LIB"] = out

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hanabizero_amd import _lib  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def main():
    N = 4096
    cfg = make_config("Hanabi-Full", simulations=50, stack=4)
    eng = bench.build_engine(cfg, torch.float16, "cuda")
    actor = SelfPlayActor(cfg, eng, num_envs=N, rank=0, seed=1, use_graph=True)
    lib = _lib.lib
    lib.hz_tail_profile_read.argtypes = [C.c_void_p]
    names = {0: ["start", "draw", "select_action", "history", "step", "observe", "end"],
             1: ["start", "slots", "flush", "reset", "observe", "heads", "end"]}
    for move in range(24):
        actor.step()
        torch.cuda.synchronize()
        if move < 20:
            continue
        prof = np.zeros(2 * 16 * 8 * 8, np.uint64)
        lib.hz_tail_profile_read(prof.ctypes.data_as(C.c_void_p))
        p = prof.astype(np.int64).reshape(2, 16, 8, 8)
        for k, kname in ((0, "k_move_tail_a"), (1, "k_move_tail_b")):
            t0 = p[k, :, :4, 0].min()
            print("move %d %s (x 10 ns from the first sampled workgroup's start); owner waves:" % (move, kname))
            for kind in (1, 2):
                rows = [p[k, g, w] for g in range(16) for w in range(4) if p[k, g, w, 7] == kind]
                if not rows:
                    continue
                r = np.array(rows)
                rel = r[:, :7] - r[:, :1]
                what = ("no deal" if kind == 1 else "deal") if k == 0 else ("game goes on" if kind == 1 else "game ended")
                print("   %-13s n=%3d  start at %5.0f..%5.0f | " % (what, len(rows), (r[:, 0] - t0).min(), (r[:, 0] - t0).max()) +
                      "  ".join("%s %5.0f" % (names[k][i], rel[:, i].mean()) for i in range(1, 7)) + "  (max end %5.0f)" % rel[:, 6].max())
            last = p[k, :, :4, 6].max()
            print("   sampled span: %d" % (last - t0))


if __name__ == "__main__":
    main()
