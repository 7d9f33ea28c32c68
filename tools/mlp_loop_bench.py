#!/usr/bin/env python3
"""tools/mlp_loop_bench.py -- the product's fused inference in a loop inside ONE launch (tools/mlp_loop.hip): bytes of the
weight stream per shader cycle and per second, for the real chain and for synthetic chains, next to tools/l2_stream_bench.hip's
ceiling for the same stream.  Extra -D flags on the command line go to hipcc (a scratch build with a switch of one's own).
  python tools/mlp_loop_bench.py [N=4096] [-D...]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import bench
    from hanabizero_amd._lib import MlpHeader
    from hanabizero_amd.config import make_config
    from hanabizero_amd.model import FusedRecurrent
    from mlp_chain_probe import uniform_chain
    N = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 4096
    flags = [f for f in sys.argv[1:] if f.startswith("-D")]
    src = os.path.join(ROOT, "hanabizero_amd", "csrc")
    out = os.path.join(ROOT, "gpurun_out", "libmlp_loop.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w",
                           "-I" + src, "-I" + os.path.join(ROOT, "include"), "-o", out, os.path.join(ROOT, "tools", "mlp_loop.hip")] + flags)
    lib = C.CDLL(out)
    V, I = C.c_void_p, C.c_int
    lib.hz_mlp_loop.argtypes = [C.POINTER(MlpHeader), V, V, V, V, V, V, I, I, I, V, V]
    cfg = make_config("Hanabi-Full", simulations=50, stack=4)
    eng = bench.build_engine(cfg, torch.bfloat16, "cuda")
    net = eng._net
    act = torch.randint(0, eng.A, (N,), device="cuda", dtype=torch.int32)
    hout = torch.empty(N, eng.H, dtype=torch.bfloat16, device="cuda")
    stamps = torch.zeros(64, dtype=torch.int64, device="cuda")
    iters = 49
    print("flags: %s" % (" ".join(flags) or "(none)"))
    for mt in (16, 32):
        n = 256 * mt
        if n > N:
            continue
        for name, ch in (("real recurrent inference", FusedRecurrent(net, eng, 16, 2)), ("6 uniform 512x512 layers", uniform_chain(eng, 16, 2, 6)),
                         ("3 uniform 512x512 layers", uniform_chain(eng, 16, 2, 3))):
            def launch():
                rc = lib.hz_mlp_loop(C.byref(ch.header), ch.jobs.data_ptr(), ch.weights.data_ptr(), ch.biases.data_ptr(), ch.act_table.data_ptr(),
                                     act.data_ptr(), hout.data_ptr(), n, mt, iters, stamps.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            best = 1e9
            for rep in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch()
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    best = min(best, e0.elapsed_time(e1))
            st = stamps.cpu().numpy().reshape(-1, 2)[: n // mt // 64]
            ticks, real = float(st[:, 0].mean()), float(st[:, 1].mean())
            wb = ch.weight_bytes_per_wg
            print("%2d rows/WG  %-28s %8.3f ms / %d inferences = %6.2f us each  %6.1f GB/s per CU | %7.0f shader cycles each  %5.1f B/cycle | shader clock %.2f GHz" % (
                mt, name, best, iters, best * 1e3 / iters, wb * iters / best / 1e6, ticks / iters, wb * iters / ticks, ticks / real * 0.1), flush=True)


if __name__ == "__main__":
    main()
