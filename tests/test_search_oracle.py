"""GPU: the BENCHED search -- `hz_search_run`, all simulations of a move in one persistent kernel (k_search / k_search_half /
k_search_turn with the fused fp16 MFMA inference) -- against the ORACLE tree directly, at BASELINE.json's sizes.

Reference: /root/reference/core/mcts.py:24-57 (the simulation loop), core/ctree/cnode.cpp:337-441 (backup, descent).
The oracle side (tests/oracle_replay.py) makes every descent and backup on the CPU with oracle/tree_oracle.c and takes from the
product only the stand-alone recurrent inference per simulation; nothing of the HIP tree code is on that side.  Equal means
equal bits: visit counts, root values, greedy trajectories, min-max statistics, last path lengths and every plane of the
hidden-state pool.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("game,N,net,moves", [
    ("Hanabi-Full", 4096, "random", 3),     # BASELINE's metric configuration: 256 workgroups x 16 trees (k_search)
    ("Hanabi-Full", 4096, "sharp", 2),      # ... with a concentrated policy: paths of 12-35 nodes, predicted-line descent
    ("Hanabi-Full", 8192, "random", 1),     # configs[2]: 32 trees per workgroup, two per wave side by side (k_search_half)
    ("Hanabi-Full", 8192, "sharp", 2),
    ("Hanabi-Full-5p", 512, "random", 2),   # A = 48: 16 trees per workgroup
    ("Hanabi-Full-5p", 4099, "sharp", 1),   # ... and more than 16 per compute unit: two per wave in turn (k_search_turn); ragged tail
    ("Hanabi-Small", 1000, "random", 4)])   # configs[1]'s game (the two-layer heads' job table)
def test_benched_search_kernel_equals_oracle_tree(game, N, net, moves):
    from hanabizero_amd import cytree
    from hanabizero_amd._lib import poll_giveups
    from hanabizero_amd.mcts import MCTS
    from oracle.cport import OracleTree
    from tests.oracle_replay import bits, oracle_search
    from tests.test_selfplay import make
    sims = 50
    giveups_before = poll_giveups()
    cfg, eng, actor = make(game, N, sims, 4, torch.float16, use_graph=False, seed=31, peaked="sharp" if net == "sharp" else False)
    assert eng.fused is not None and eng.fused.header.dtype == 2
    A = cfg.action_space_size
    for _ in range(moves):  # positions a few moves into the games (legal masks and windows that differ between the envs)
        actor.step()
    actor._draw()
    _, logits0, hidden0 = actor.root_inference()
    noise, legal = actor.noise.clone(), actor.legal.clone()
    torch.cuda.synchronize()
    # ---- oracle: CPU tree, one stand-alone inference launch per simulation
    tree = OracleTree(N, A, sims, seed=7, value_delta_max=cfg.value_delta_max, tree_id_base=1000)
    tree.prepare(cfg.root_exploration_fraction, noise.cpu().numpy(), np.zeros(N, np.float32), logits0.cpu().numpy(), legal.cpu().numpy())
    pool_o = oracle_search(cfg, eng, tree, hidden0, sims)
    want = dict(dist=tree.distributions(), values=tree.values(), traj=tree.trajectories(), minmax=tree.minmax(), plen=tree.path_len())
    assert int(want["dist"].sum()) == N * (sims - 1)
    if net == "sharp":
        assert int(want["plen"].max()) > 8, int(want["plen"].max())
    # ---- product: ONE hz_search_run launch, the kernel shape of the library's choice; with and without the predicted-line descent
    for lines in (True, False):
        roots = cytree.Roots(N, A, sims, tie_seed=7, tree_id_base=1000)
        roots.set_predicted_lines(lines)
        roots.prepare(cfg.root_exploration_fraction, noise, torch.zeros(N, device="cuda"), logits0, legal)
        pool = torch.zeros(sims, N, eng.H, dtype=eng.dtype, device="cuda")
        MCTS(cfg, persistent=True).run_multi(roots, eng, hidden0, pool=pool)
        torch.cuda.synchronize()
        assert roots._sim == sims - 1, "the persistent kernel did not run (fell back to the launch-per-phase search)"
        assert np.array_equal(roots.distributions_tensor().cpu().numpy(), want["dist"]), (lines, "visit counts")
        assert np.array_equal(bits(roots.values_tensor().cpu().numpy()), bits(want["values"])), (lines, "root values")
        assert np.array_equal(roots.trajectories_tensor().cpu().numpy(), want["traj"]), (lines, "trajectories")
        mn, mx = roots.minmax_tensors()
        assert np.array_equal(bits(mn.cpu().numpy()), bits(want["minmax"][0])) and np.array_equal(bits(mx.cpu().numpy()), bits(want["minmax"][1]))
        assert np.array_equal(roots.path_len_tensor().cpu().numpy(), want["plen"]), (lines, "last path lengths")
        assert torch.equal(bits(pool), bits(pool_o)), (lines, "hidden-state pool")
        del roots
    assert poll_giveups() == giveups_before, "a wave gave up waiting for an arrival counter (include/hz_mlp.h)"


@pytest.mark.parametrize("game,N,net,moves", [("Hanabi-Full", 4096, "random", 2), ("Hanabi-Full", 1500, "sharp", 2), ("Hanabi-Small", 1000, "random", 3),
                                               ("Hanabi-Full-5p", 700, "random", 2)])   # (A = 48: the widest tree rows beside the two-plane image)
def test_fp16_pair_engine_search_equals_oracle_tree(game, N, net, moves):
    """The engine inside the contract's 1e-3 (InferenceEngine(dtype=float32, fused="fp16x2"): fp32 pool, recurrent inference =
    the MFMA kernel's HZ_F16X2 build) searches in ONE persistent kernel (k_search_pairs, 16 trees per workgroup) or launch by
    launch -- descent / backup kernels around the stand-alone inference kernel, two launches per simulation -- and either way
    must leave what the oracle tree leaves when the stand-alone inference kernel evaluates the oracle's own leaves: visit counts,
    root values, trajectories, min-max statistics, pool planes, bit for bit."""
    from hanabizero_amd import cytree
    from hanabizero_amd.mcts import MCTS
    from oracle.cport import OracleTree
    from tests.oracle_replay import bits, oracle_search
    from tests.test_selfplay import make
    sims = 50
    cfg, eng, actor = make(game, N, sims, 4, torch.float32, use_graph=False, seed=31, peaked="sharp" if net == "sharp" else False, fused="fp16x2")
    assert eng.fused is not None and eng.fused.header.dtype == 3 and eng.fused_shape(16, 2).header.dtype == 3
    A = cfg.action_space_size
    for _ in range(moves):
        actor.step()
    actor._draw()
    _, logits0, hidden0 = actor.root_inference()
    noise, legal = actor.noise.clone(), actor.legal.clone()
    torch.cuda.synchronize()
    tree = OracleTree(N, A, sims, seed=7, value_delta_max=cfg.value_delta_max, tree_id_base=1000)
    tree.prepare(cfg.root_exploration_fraction, noise.cpu().numpy(), np.zeros(N, np.float32), logits0.cpu().numpy(), legal.cpu().numpy())
    pool_o = oracle_search(cfg, eng, tree, hidden0, sims)
    for persistent in (True, False):
        roots = cytree.Roots(N, A, sims, tie_seed=7, tree_id_base=1000)
        roots.prepare(cfg.root_exploration_fraction, noise, torch.zeros(N, device="cuda"), logits0, legal)
        pool = torch.zeros(sims, N, eng.H, dtype=torch.float32, device="cuda")
        MCTS(cfg, persistent=persistent).run_multi(roots, eng, hidden0, pool=pool)
        torch.cuda.synchronize()
        assert roots._sim == sims - 1
        assert np.array_equal(roots.distributions_tensor().cpu().numpy(), tree.distributions()), persistent
        assert np.array_equal(bits(roots.values_tensor().cpu().numpy()), bits(tree.values())), persistent
        assert np.array_equal(roots.trajectories_tensor().cpu().numpy(), tree.trajectories()), persistent
        mn, mx = roots.minmax_tensors()
        assert np.array_equal(bits(mn.cpu().numpy()), bits(tree.minmax()[0])) and np.array_equal(bits(mx.cpu().numpy()), bits(tree.minmax()[1]))
        assert torch.equal(bits(pool), bits(pool_o)), persistent
        del roots
