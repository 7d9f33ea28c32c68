#!/usr/bin/env python3
"""tools/train_block_bench.py -- the learner's Linear + BatchNorm + ReLU block, forward + backward: PyTorch autograd under bf16
autocast against the fused block (include/hz_train.h), as hipGraph replays of 20 blocks in a chain (what a learner step is made of);
and the two hand-written kernels alone.  Prints microseconds per block."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from hanabizero_amd._lib import check, lib  # noqa: E402
from hanabizero_amd.fused_train import _Block, _LinBNAct  # noqa: E402


def graph_time(fn, reps=50):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    out = {}
    for B, Cn in ((256, 512), (256, 1024), (256, 256)):
        L = 20
        lins = [nn.Linear(Cn, Cn).cuda() for _ in range(L)]
        bns = [nn.BatchNorm1d(Cn).cuda() for _ in range(L)]
        for m in lins + bns:
            for p in m.parameters():
                p.grad = torch.zeros_like(p)
        x0 = torch.randn(B, Cn, device="cuda")
        anchor = torch.zeros(1, device="cuda", requires_grad=True)

        def eager():
            with torch.autocast("cuda", dtype=torch.bfloat16):
                x = x0 + anchor
                for lin, bn in zip(lins, bns):
                    x = torch.relu(bn(lin(x)))
            x.float().sum().backward()
        blocks = [_Block(lin, bn, torch.bfloat16, 1) for lin, bn in zip(lins, bns)]
        for b in blocks:
            b.w16.copy_(b.lin.weight.detach())
            b.w16t.copy_(b.lin.weight.detach().t())
        x16 = x0.to(torch.bfloat16)

        def fused():
            x, xt = x16, None
            for k, b in enumerate(blocks):
                x, xt = _LinBNAct.apply(x, xt, None, b, True, anchor if k == 0 else None)
            x.float().sum().backward()
        te, tf = graph_time(eager), graph_time(fused)
        # the two kernels alone
        y = torch.randn(B, Cn, device="cuda").to(torch.bfloat16)
        o, dx = torch.empty_like(y), torch.empty_like(y)
        st8 = torch.empty(2, Cn, device="cuda")
        bn = bns[0]
        s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def kf():
            for _ in range(20):
                check(lib.hz_bn_act_forward(y.data_ptr(), Cn, None, 0, o.data_ptr(), Cn, B, Cn, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                            bn.running_var.data_ptr(), 0.1, 1e-5, st8[0].data_ptr(), st8[1].data_ptr(), 1, 1, s()), "f")

        def kb():
            for _ in range(20):
                check(lib.hz_bn_act_backward(y.data_ptr(), Cn, o.data_ptr(), Cn, y.data_ptr(), Cn, dx.data_ptr(), Cn, None, 0, None, 0, B, Cn, bn.weight.data_ptr(),
                                             st8[0].data_ptr(), st8[1].data_ptr(), bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr(), 1, 1, s()), "b")
        # the GEMMs alone, hot and cold: hipBLASLt through torch against hz_gemm_nt (16-bit store; fp32 accumulate with the short reduction of a
        # weight gradient).  "cold": 24 different weight matrices in turn (what a step does: a layer's weights were last touched a step ago)
        a16 = torch.randn(B, Cn, device="cuda").bfloat16()
        ws = [torch.randn(Cn, Cn, device="cuda").bfloat16() for _ in range(24)]
        o16, g32 = torch.empty(B, Cn, device="cuda", dtype=torch.bfloat16), torch.zeros(Cn, Cn, device="cuda")
        at = a16.t().contiguous()
        st_ = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def hz(xx, ww, rows, cols, k, oo, epi):
            check(lib.hz_gemm_nt(xx.data_ptr(), xx.stride(0), ww.data_ptr(), ww.stride(0), None, rows, cols, k, oo.data_ptr(), oo.stride(0), epi, None, 1, st_()), "g")

        def t_mm():
            for i in range(24):
                torch.mm(a16, ws[0].t(), out=o16)

        def h_mm():
            for i in range(24):
                hz(a16, ws[0], B, Cn, Cn, o16, 0)

        def t_mm_cold():
            for i in range(24):
                torch.mm(a16, ws[i].t(), out=o16)

        def h_mm_cold():
            for i in range(24):
                hz(a16, ws[i], B, Cn, Cn, o16, 0)

        def t_dw():
            for _ in range(24):
                torch.addmm(g32, at, a16, out_dtype=torch.float32, out=g32)

        def h_dw():
            for _ in range(24):
                hz(at, at, Cn, Cn, B, g32, 2)
        gem = {"torch_mm_hot_us": 1e6 * graph_time(t_mm) / 24, "hz_gemm_nt_store_hot_us": 1e6 * graph_time(h_mm) / 24,
               "torch_mm_24_weights_us": 1e6 * graph_time(t_mm_cold) / 24, "hz_gemm_nt_store_24_weights_us": 1e6 * graph_time(h_mm_cold) / 24,
               "torch_addmm_fp32_dw_us": 1e6 * graph_time(t_dw) / 24, "hz_gemm_nt_acc32_dw_us": 1e6 * graph_time(h_dw) / 24}
        out["%dx%d" % (B, Cn)] = {**gem, "autograd_autocast_us_per_block_fwd_bwd": 1e6 * te / L, "fused_us_per_block_fwd_bwd": 1e6 * tf / L,
                                  "hz_bn_act_forward_us": 1e6 * graph_time(kf) / 20, "hz_bn_act_backward_us": 1e6 * graph_time(kb) / 20}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
