"""Test infrastructure: MCTS.run_multi (core/mcts.py:11-57) replayed on the CPU -- the ORACLE tree (oracle/tree_oracle.c:
cnode.cpp's arithmetic restated) makes every descent and every backup; the only thing taken from the product is the nets'
recurrent inference, called once per simulation over the whole batch exactly as core/mcts.py:38-42 calls the model:

  * 16-bit engines: the stand-alone fused MFMA kernel (`eng.fused` = hz_mlp_recurrent, include/hz_mlp.h) -- the code the
    persistent search kernel inlines -- fed with the oracle's (hidden_state_index_x, last_action) per tree;
  * fp32 engines: the hipBLASLt GEMM chain (`eng.recurrent_heads`) + the HIP scalar transform.

So a search by `hz_search_run` (ONE persistent kernel: k_search / k_search_half / k_search_turn) and this replay share
nothing of the tree: every visit count, root value, trajectory and pool row that comes out equal was computed twice.
"""
import numpy as np
import torch


def oracle_search(cfg, eng, tree, hidden0, sims, pool=None):
    """Drives `tree` (oracle.cport.OracleTree, prepared) through sims - 1 simulations (core/mcts.py:24-26).  Returns the
    hidden-state pool [sims, N, H] it built (plane 0 = hidden0), as the persistent kernel leaves it in HBM."""
    N, A = tree.N, tree.A
    dev = hidden0.device
    if pool is None:
        pool = torch.zeros(sims, N, eng.H, dtype=eng.dtype, device=dev)
    pool[0].copy_(hidden0)
    fused = getattr(eng, "fused", None)
    rew = torch.empty(N, dtype=torch.float32, device=dev)
    val = torch.empty(N, dtype=torch.float32, device=dev)
    pol = torch.empty(N, A, dtype=torch.float32, device=dev)
    for sim in range(sims - 1):
        ix, iy, la = tree.traverse(sim, cfg.pb_c_base, cfg.pb_c_init, cfg.discount)
        assert (iy == np.arange(N)).all() and (ix <= sim).all() and (ix >= 0).all()
        if fused is not None:
            fused(pool, torch.from_numpy(ix).to(dev), torch.from_numpy(la).to(dev), pool[sim + 1], rew, val, pol)
            r, v, lg = rew.cpu().numpy(), val.cpu().numpy(), pol.cpu().numpy()
        else:
            hid = pool[torch.from_numpy(ix).long().to(dev), torch.arange(N, device=dev)]
            net_in = torch.zeros(N, eng.H + eng.onehot_cols, dtype=eng.dtype, device=dev)
            net_in[:, :eng.H] = hid
            net_in[torch.arange(N, device=dev), eng.H + torch.from_numpy(la).long().to(dev)] = 1
            r_log, v_log, p_log = eng.recurrent_heads(net_in, pool[sim + 1])
            r, v = eng.support_to_scalar(r_log).cpu().numpy(), eng.support_to_scalar(v_log).cpu().numpy()
            lg = torch.nan_to_num(p_log[:, :A].float(), nan=0.0, posinf=float("inf"), neginf=float("-inf")).cpu().numpy()  # core/mcts.py:48-49
        tree.backprop(sim + 1, cfg.discount, r, v, lg)
    return pool


def bits(t):
    """Bit patterns (a deep fp16 chain of random nets may hold inf / NaN, which compare unequal as numbers)."""
    if isinstance(t, np.ndarray):
        return t.view({2: np.uint16, 4: np.uint32, 8: np.uint64}[t.dtype.itemsize]) if t.dtype.kind == "f" else t
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.view(torch.int16)
    return t.view(torch.int32) if t.dtype == torch.float32 else t
