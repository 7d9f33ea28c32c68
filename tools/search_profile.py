#!/usr/bin/env python3
"""tools/search_profile.py -- diagnostic build of the persistent search kernel (-DHZ_SEARCH_PROFILE): where the waves of
workgroup 100 spend their cycles over one move's search (tree phases, MFMA inference phases, barrier waits).
Builds a scratch copy of the library (HANABIZERO_HIP_LIB points the loader at it).  Never quote run times of it."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from _build import build  # noqa: E402
out = build("libsearch_prof.so", {"hz_search.hip": ["-DHZ_SEARCH_PROFILE", "-DHZ_TREE_PROFILE"] + sys.argv[2:]})
os.environ["HANABIZERO_HIP_LIB"] = out

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hanabizero_amd import _lib  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cfg = make_config("Hanabi-Full", simulations=50, stack=4)
    eng = bench.build_engine(cfg, torch.bfloat16, "cuda", net=os.environ.get("NET", "random"))  # NET=sharp: deep paths
    actor = SelfPlayActor(cfg, eng, num_envs=N, rank=0, seed=1, use_graph=False,
                          predicted_lines={"on": True, "off": False}.get(os.environ.get("LINES", "auto"), "auto"))  # LINES=on|off
    actor.mcts.rows_per_workgroup = int(os.environ.get("ROWS", "0"))  # ROWS=16|32 forces
    for _ in range(3):
        actor.step()
    torch.cuda.synchronize()
    lib = _lib.lib
    lib.hz_search_profile_read.argtypes = [C.c_void_p]
    prof = np.zeros(64, np.uint64)
    lib.hz_search_profile_read(prof.ctypes.data_as(C.c_void_p))
    p = prof.astype(np.int64).reshape(16, 4)
    print("k_search, workgroup 100, one move (%d simulations): s_memtime ticks per wave" % (cfg.num_simulations - 1))
    for w in range(16):
        print("  wave %2d: tree phases %8d | wait for the other trees %8d | inference %8d | wait after inference %7d | total %8d"
              % (w, p[w, 0], p[w, 1], p[w, 2], p[w, 3], p[w].sum()))
    print("  per simulation (mean over waves): tree %.0f, wait %.0f, inference %.0f, wait %.0f"
          % tuple(p.mean(0) / (cfg.num_simulations - 1)))
    # the tree phase of the wave that owns tree 400, last simulation with a descent (stamps of hz_tree_dev.h)
    lib.hz_tree_profile_read.argtypes = [C.c_void_p]
    tp = np.zeros(16, np.uint64)
    lib.hz_tree_profile_read(tp.ctypes.data_as(C.c_void_p))
    tp = tp.astype(np.int64)
    names = ["start", "expand done", "leaf value/reward", "backup loop", "min/max", "fence"] + ["level %d" % d for d in range(1, 8)] + ["end"]
    print("    heads' read-out        +%6d" % (tp[0] - tp[14]))
    prev = tp[0]
    for i, nme in enumerate(names):
        if tp[i] == 0 or tp[i] < tp[0]:
            continue
        print("    %-20s +%6d  (at %6d)" % (nme, tp[i] - prev, tp[i] - tp[0]))
        prev = tp[i]


if __name__ == "__main__":
    main()
