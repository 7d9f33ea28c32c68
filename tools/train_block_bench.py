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
        x16 = x0.to(torch.bfloat16)

        def fused():
            x = x16
            for k, b in enumerate(blocks):
                x = _LinBNAct.apply(x, None, b, True, anchor if k == 0 else None)
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
                check(lib.hz_bn_act_backward(y.data_ptr(), Cn, o.data_ptr(), Cn, y.data_ptr(), Cn, dx.data_ptr(), Cn, None, 0, B, Cn, bn.weight.data_ptr(),
                                             st8[0].data_ptr(), st8[1].data_ptr(), bn.weight.grad.data_ptr(), bn.bias.grad.data_ptr(), 1, 1, s()), "b")
        out["%dx%d" % (B, Cn)] = {"autograd_autocast_us_per_block_fwd_bwd": 1e6 * te / L, "fused_us_per_block_fwd_bwd": 1e6 * tf / L,
                                  "hz_bn_act_forward_us": 1e6 * graph_time(kf) / 20, "hz_bn_act_backward_us": 1e6 * graph_time(kb) / 20}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
