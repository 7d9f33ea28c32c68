#!/usr/bin/env python3
"""tools/level_profile.py [envs] [random|sharp[:scale]] -- diagnostic build (-DHZ_TREE_PROFILE): where a LEVEL of the descent
spends its cycles inside the persistent search kernel, summed over every level of one move's search for the wave that owns
tree HZ_TREE_PROFILE_TREE (pass -DHZ_TREE_PROFILE_TREE=n after the two arguments for another wave).  Never quote run times."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _build import build  # noqa: E402
out = build("liblevel_prof.so", {"hz_search.hip": ["-DHZ_TREE_PROFILE"] + sys.argv[3:]})
os.environ["HANABIZERO_HIP_LIB"] = out

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hanabizero_amd import _lib  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def read(lib):
    t = np.zeros(64 * 32, np.uint64)
    lib.hz_tree_level_profile_read(t.ctypes.data_as(C.c_void_p))
    return t.astype(np.int64).reshape(64, 32).sum(0)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    net = sys.argv[2] if len(sys.argv) > 2 else "random"
    cfg = make_config("Hanabi-Full", simulations=50, stack=4)
    eng = bench.build_engine(cfg, torch.float16, "cuda", net=net)
    actor = SelfPlayActor(cfg, eng, num_envs=N, rank=0, seed=1, use_graph=False, predicted_lines=True)
    lib = _lib.lib
    lib.hz_tree_level_profile_read.argtypes = [C.c_void_p]
    for _ in range(5):
        actor.step()
    torch.cuda.synchronize()
    a = read(lib)
    M = 4
    for _ in range(M):
        actor.step()
    torch.cuda.synchronize()
    d = read(lib) - a
    lv = max(int(d[8]), 1)
    names = ["wait for records", "ordered sum + mean q", "scores + max", "ties, action", "fence (level 0), stores", "child lanes, next request"]
    print("net %s, %d envs: %d levels in %d moves; cycles per level: %s | total %.0f"
          % (net, N, lv, M, " | ".join("%s %.0f" % (n, d[i] / lv) for i, n in enumerate(names)), d[:8].sum() / lv))
    print("  %d descents, %d of them began as a replay: %.1f levels and %.0f cycles per replay; ordinary levels %d"
          % (d[14], d[11], d[12] / max(int(d[11]), 1), d[13] / max(int(d[11]), 1), lv))
    pn = ["the line (pointer doubling)", "records + q cache", "ordered sums", "mean q chain", "scores + max", "selection", "commit + hand-over"]
    np_ = max(int(d[24]), 1)
    print("  %d passes; cycles per pass: %s | total %.0f" % (d[24], " | ".join("%s %.0f" % (n, d[16 + i] / np_) for i, n in enumerate(pn)), d[16:23].sum() / np_))
    print("  whole descents (of >= 2 levels and >= the minimum depth): %d levels (%d of them inside replays), %.0f cycles per level"
          % (d[10], d[15], d[9] / max(int(d[10]), 1)))

if __name__ == "__main__":
    main()
