#!/usr/bin/env python3
"""tools/pairing_probe.py -- would pairing trees by depth pay in the side-by-side 32-tree search kernel?

k_search_half walks two trees per wave in its halves: a wave's descent costs max(depth_a, depth_b) levels, and the four waves of
a SIMD share its issue port, so a workgroup's tree phase lasts about max over SIMDs of the sum of its waves' costs.  This probe
plays a few moves with the sharp-policy net, records EVERY simulation's path length of every tree (launch-per-phase search) for
two consecutive moves, and evaluates that cost model for: the identity assignment (tree 32 b + 2 w + h), pairs sorted by the
PREVIOUS move's total path length (what a kernel can know at launch), by THIS move's (oracle, static), and re-sorted before
every simulation by the previous simulation's length (dynamic)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from hanabizero_amd import cytree  # noqa: E402
from hanabizero_amd.config import make_config  # noqa: E402
from hanabizero_amd.selfplay import SelfPlayActor  # noqa: E402


def per_sim_lengths(actor, cfg, eng):
    N, S, A = actor.N, actor.S, actor.A
    _, logits0, hidden0 = actor.root_inference()
    roots = cytree.Roots(N, A, S, tie_seed=1, tree_id_base=0)
    roots.prepare(cfg.root_exploration_fraction, actor.noise, torch.zeros(N, device="cuda"), logits0, actor.legal)
    roots.set_params(cfg.pb_c_base, cfg.pb_c_init, cfg.discount, cfg.value_delta_max)
    pool = torch.zeros(S, N, eng.H, dtype=eng.dtype, device="cuda")
    pool[0].copy_(hidden0)
    rew, val, pol = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, A, device="cuda")
    lens = []
    ix, _, la = roots.traverse_tensors()
    for sim in range(S - 1):
        lens.append(roots.path_len_tensor().cpu().numpy().copy())
        eng.fused(pool, ix, la, pool[sim + 1], rew, val, pol)
        if sim < S - 2:
            ix, _, la = roots.backprop_traverse_tensors(sim + 1, rew, val, pol)
        else:
            roots.backprop_tensors(sim + 1, rew, val, pol)
    return np.stack(lens)  # [S-1, N] nodes on the path (edges + 1)


def level_cost(n):
    """cycles of one half's descent of n nodes with the predicted-line passes: passes of 8 levels at ~4 k cycles once a tree is deep
    (n >= 6), ~2.2 k per level otherwise."""
    lv = np.maximum(n - 1, 0)
    return np.where(lv >= 6, np.ceil(lv / 8.0) * 4000.0 + 1500.0, lv * 2200.0)


def phase_cost(lens, perm):
    """lens [S-1, N]; perm [N/32, 32]: position 2 w + h of workgroup b holds tree perm[b, 2 w + h].  Sum over simulations of the
    workgroups' mean (max over SIMDs of the sum over its 4 waves of max over the 2 halves)."""
    c = level_cost(lens)[:, perm]                       # [S-1, B, 32]
    wave = c.reshape(c.shape[0], c.shape[1], 16, 2).max(3)
    simd = wave.reshape(c.shape[0], c.shape[1], 4, 4).sum(2)   # waves w, w + 4, w + 8, w + 12 share SIMD w % 4
    return float(simd.max(2).mean(1).sum())


def sorted_perm(key, N):
    """Per workgroup of 32: trees sorted by key (descending), adjacent pairs into one wave, pairs dealt to the SIMDs in snake order."""
    B = N // 32
    perm = np.zeros((B, 32), np.int64)
    snake = [0, 1, 2, 3, 3, 2, 1, 0, 0, 1, 2, 3, 3, 2, 1, 0]
    for b in range(B):
        idx = np.arange(32 * b, 32 * b + 32)
        order = idx[np.argsort(-key[idx], kind="stable")]
        slot = [0, 0, 0, 0]
        for r in range(16):
            s = snake[r]
            w = s + 4 * slot[s]
            slot[s] += 1
            perm[b, 2 * w], perm[b, 2 * w + 1] = order[2 * r], order[2 * r + 1]
    return perm


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    net = sys.argv[2] if len(sys.argv) > 2 else "sharp"
    cfg = make_config("Hanabi-Full", simulations=50, stack=4, p_mcts_num=N)
    eng = bench.build_engine(cfg, torch.float16, torch.device("cuda", 0), net=net)
    actor = SelfPlayActor(cfg, eng, N, seed=0, use_graph=False)
    for _ in range(6):
        actor.step()
    torch.cuda.synchronize()
    out = {"envs": N, "net": net, "moves": []}
    prev = None
    for move in range(4):
        lens = per_sim_lengths(actor, cfg, eng)
        ident = np.arange(N).reshape(N // 32, 32)
        entry = {"mean_nodes": float(lens.mean()), "max_nodes": int(lens.max()), "identity": phase_cost(lens, ident),
                 "oracle_static_this_move": phase_cost(lens, sorted_perm(lens.sum(0), N))}
        if prev is not None:
            entry["by_previous_move_total"] = phase_cost(lens, sorted_perm(prev.sum(0), N))
            entry["by_previous_move_last_sim"] = phase_cost(lens, sorted_perm(prev[-1], N))
            entry["corr_total_prev_this"] = float(np.corrcoef(prev.sum(0), lens.sum(0))[0, 1])
        dyn = 0.0
        for s in range(lens.shape[0]):
            key = lens[s - 1] if s else np.zeros(N)
            dyn += phase_cost(lens[s:s + 1], sorted_perm(key, N))
        entry["dynamic_by_previous_simulation"] = dyn
        for k in list(entry):
            if k not in ("mean_nodes", "max_nodes", "identity", "corr_total_prev_this"):
                entry[k + "_vs_identity"] = entry[k] / entry["identity"]
        out["moves"].append(entry)
        if len(sys.argv) > 3:
            np.save(sys.argv[3] + "_move%d.npy" % move, lens.astype(np.int8))
        prev = lens
        actor.step()
        torch.cuda.synchronize()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
