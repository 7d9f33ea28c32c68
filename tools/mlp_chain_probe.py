#!/usr/bin/env python3
"""tools/mlp_chain_probe.py -- the fused MLP kernel (hz_mlp_recurrent) on SYNTHETIC layer chains: the job table is data, so the
same kernel can be timed on chains that isolate what the real chain's structure costs -- uniform 512 x 512 layers (every wave
busy, one barrier per layer), the dynamics part alone, the head stages alone -- against tools/l2_stream_bench.hip's ceiling for
the same number of weight bytes.  Prints us per launch and GB/s per CU of the weight stream."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def uniform_chain(eng, waves, tiles, n, seed=0):
    """n layers of H x H (every wave busy, one barrier per layer), ping-pong between two image regions."""
    from hanabizero_amd.model import _FusedChain
    H = eng.H
    g = torch.Generator().manual_seed(seed)
    ch = _FusedChain(eng, waves, tiles)
    X, Y = 0, H
    for k in range(n):
        w, b = torch.randn(H, H, generator=g) / H ** 0.5, torch.randn(H, generator=g) * 0.1
        ch.add_dense(w, b, H, X if k % 2 == 0 else Y, Y if k % 2 == 0 else X, relu=True, barrier=k > 0, store_hidden=(k == n - 1))
    ch._finish(3 * H, in_width=H, hidden=H, state_off=0, hidden_off=Y if n % 2 == 0 else X, off_r=2 * H, off_v=2 * H + 256, off_p=2 * H + 512)
    return ch


def main():
    import bench
    from hanabizero_amd._lib import check, lib
    from hanabizero_amd.config import make_config
    from hanabizero_amd.model import FusedRecurrent, _FusedChain, _fold
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cfg = make_config("Hanabi-Full", simulations=50, stack=4)
    eng = bench.build_engine(cfg, torch.bfloat16, "cuda")
    net = eng._net
    H, A, h = eng.H, eng.A, eng.h
    g = torch.Generator().manual_seed(0)
    rnd = lambda o, i: (torch.randn(o, i, generator=g) / i ** 0.5, torch.randn(o, generator=g) * 0.1)

    Uniform = lambda engine, waves, tiles, n: uniform_chain(engine, waves, tiles, n)

    class Dynamics(_FusedChain):  # the three dynamics layers of the real net (action row + residual) + hidden store
        def __init__(self, engine, waves, tiles):
            super().__init__(engine, waves, tiles)
            dyn = net._dynamics_state
            w1, b1 = _fold(dyn.fc1, dyn.bn1)
            X, Y1, Y0 = 0, H, 2 * H
            self.add_dense(w1[:, :H], b1, H, X, Y0, relu=True, barrier=False, act_w=w1[:, H:])
            self.add_dense(*_fold(dyn.fc2, dyn.bn2), H, Y0, Y1, relu=True)
            self.add_dense(*_fold(dyn.fc3, dyn.bn3), H, Y1, Y0, relu=True, res_off=X)
            w, b = rnd(H, H)
            self.add_dense(w, b, H, Y0, X, relu=True, store_hidden=True)
            self._finish(3 * H + h, in_width=H, hidden=H, state_off=0, hidden_off=Y0, off_r=0, off_v=256, off_p=512)

    def time(chain, mt):
        S = 8
        pool = torch.rand(S, N, H, device="cuda").to(torch.bfloat16)
        ix = torch.randint(0, S, (N,), device="cuda", dtype=torch.int32)
        act = torch.randint(0, A, (N,), device="cuda", dtype=torch.int32)
        hout = torch.empty(N, H, dtype=torch.bfloat16, device="cuda")
        r, v, p = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, A, device="cuda")

        def launch():
            check(lib.hz_mlp_recurrent(C.byref(chain.header), chain.jobs.data_ptr(), chain.weights.data_ptr(), chain.biases.data_ptr(),
                                       chain.act_table.data_ptr(), pool.data_ptr(), pool.stride(1), ix.data_ptr(), pool.stride(0),
                                       act.data_ptr(), hout.data_ptr(), r.data_ptr(), v.data_ptr(), p.data_ptr(), N, mt,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hz_mlp_recurrent")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(20):
                    launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gr.replay()
            e0.record()
            for _ in range(5):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / 100

    for waves, tiles in ((16, 2), (8, 4)):
        for mt in (16, 32):
            if N // mt != 256:
                continue
            chains = [("real recurrent inference", FusedRecurrent(net, eng, waves, tiles)),
                      ("6 uniform 512x512 layers", Uniform(eng, waves, tiles, 6)),
                      ("12 uniform 512x512 layers", Uniform(eng, waves, tiles, 12)),
                      ("dynamics (3 layers) + 1 uniform", Dynamics(eng, waves, tiles))]
            for name, ch in chains:
                us = time(ch, mt)
                print("%2d x %d, %2d rows/WG, %-34s %7.2f us/launch  %6.2f MB  %6.1f GB/s per CU  (%d passes)" % (
                    waves, tiles, mt, name, us, ch.weight_bytes_per_wg / 1e6, ch.weight_bytes_per_wg / us / 1e3, ch.n_jobs), flush=True)


if __name__ == "__main__":
    main()
