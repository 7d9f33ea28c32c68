#!/usr/bin/env python3
"""tools/tree_profile.py -- diagnostic build of the tree kernels (-DHZ_TREE_PROFILE): where a wave's cycles go in
k_backprop_traverse.  Builds a scratch copy of the library with hipcc, runs a search with it (HANABIZERO_HIP_LIB points the
loader at it), prints the s_memtime stamps of the wave that owns tree 400 for the last fused launch.  Never quote run times of
this build."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _build import build  # noqa: E402
out = build("libtree_prof.so", {"hz_tree.hip": ["-DHZ_TREE_PROFILE"]})
os.environ["HANABIZERO_HIP_LIB"] = out

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hanabizero_amd import _lib  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    stop_sim = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cfg = make_config("Hanabi-Full", simulations=(stop_sim + 2 if stop_sim else 50), stack=4)
    eng = bench.build_engine(cfg, torch.bfloat16, "cuda")
    actor = SelfPlayActor(cfg, eng, num_envs=N, rank=0, seed=1, use_graph=False)
    for _ in range(3):
        actor.step()
    torch.cuda.synchronize()
    lib = _lib.lib
    lib.hz_tree_profile_read.argtypes = [C.c_void_p]
    prof = np.zeros(16, np.uint64)
    lib.hz_tree_profile_read(prof.ctypes.data_as(C.c_void_p))
    p = prof.astype(np.int64)
    names = ["start", "expand done", "leaf value/reward", "backup loop", "min/max", "fence"] + \
            ["level %d" % d for d in range(1, 8)] + ["end"]
    print("k_backprop_traverse, wave of tree 400, last fused launch of a %d-simulation search (s_memtime ticks)" % cfg.num_simulations)
    prev = p[0]
    for i, n in enumerate(names):
        if p[i] == 0 or p[i] < p[0]:
            continue
        print("  %-20s +%6d  (at %6d)" % (n, p[i] - prev, p[i] - p[0]))
        prev = p[i]


if __name__ == "__main__":
    main()
