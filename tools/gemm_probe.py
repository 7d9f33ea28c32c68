#!/usr/bin/env python3
"""tools/gemm_probe.py -- the root inference's library GEMMs (hipBLASLt through torch): time of each as the engine issues it
(bias + ReLU epilogue, [N, K] x [K, M] bf16), and what other formulations of the same product cost."""
import sys
import time

import torch


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    dt = torch.bfloat16
    for K, M in ((3168, 1024), (1024, 1024), (1024, 512)):
        x = (torch.rand(N, K, device="cuda") < 0.2).to(dt)
        w = (torch.randn(M, K, device="cuda") / K ** 0.5).to(dt)
        b = torch.randn(M, device="cuda").to(dt)
        wt = w.t().contiguous()           # [K, M] row-major
        wtt = w.t()                       # [K, M] as a view of [M, K] (TN)
        out = torch.empty(N, M, device="cuda", dtype=dt)
        r = {}
        r["addmm_activation(x, wt contiguous)"] = t(lambda: torch._addmm_activation(b, x, wt, use_gelu=False))
        r["addmm_activation(x, w.t() view)"] = t(lambda: torch._addmm_activation(b, x, wtt, use_gelu=False))
        r["addmm + relu_"] = t(lambda: torch.addmm(b, x, wt, out=out).relu_())
        r["mm only"] = t(lambda: torch.mm(x, wt, out=out))
        r["mm only (TN view)"] = t(lambda: torch.mm(x, wtt, out=out))
        r["F.linear"] = t(lambda: torch.nn.functional.linear(x, w, b))
        fl = 2.0 * N * K * M
        print("N=%d K=%d M=%d" % (N, K, M))
        for k, v in r.items():
            print("   %-38s %7.2f us  %6.0f TFLOP/s" % (k, v, fl / v / 1e6))


if __name__ == "__main__":
    main()
