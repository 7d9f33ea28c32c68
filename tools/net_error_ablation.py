#!/usr/bin/env python3
"""tools/net_error_ablation.py -- where the 16-bit engines' net error comes from, layer by layer (CPU, no GPU needed).

Emulates the fused MFMA inference of Hanabi-Full in float64 with fp16 (or bf16) rounding applied at exactly the points the
kernel rounds (weights once after the BatchNorm fold; activations at every layer's epilogue; the head logits stay fp32) and
switches those points on one at a time.  Inputs / truth: tests/golden/nets_Hanabi-Full.npz (the reference nets' fp32
outputs, 32 rows) and the wide sample of nets_Hanabi-Full_autocast.npz (256 rows, with the reference's own fp16-autocast
outputs beside it).  Error = |got - ref| / max(1, |ref|) as everywhere (tests/netgold.py).

What it shows (r03, fp16): rounding the INPUT hidden state to 16 bits -- which any engine with a 16-bit hidden-state pool
does, the reference's autocast search included -- already moves the reward scalar by 2.9e-3 (worst of 32 rows); no single
layer dominates; the head logits' own rounding (removed in r03) was 0.8e-3 of 6.8e-3; a head's first layer's weight rounding
acts on positive-mean ReLU inputs and comes out as a near-constant shift of the scalar (mean ~ 0.8 rms), whose sign and size are
a property of the weight set."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hanabizero_amd.model import _fold, inverse_scalar_transform  # noqa: E402
from tests.test_model import build  # noqa: E402

torch.set_grad_enabled(False)


def err(g, w):
    g, w = np.asarray(g, np.float64).reshape(-1), np.asarray(w, np.float64).reshape(-1)
    e = np.abs(g - w) / np.maximum(1.0, np.abs(w))
    return "%.2e/%.2e/%.2e" % (e.max(), e.mean(), np.sqrt((e * e).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    q = lambda t: t.to(dt).float()
    net, fx, sup = build("Hanabi-Full")
    ac = dict(np.load(os.path.join(ROOT, "tests", "golden", "nets_Hanabi-Full_autocast.npz")))
    dyn, rw, acn, va, rep = net._dynamics_state, net._dynamics_reward, net._prediction_actor, net._prediction_value, net._representation
    H = 512

    def lin(name, x, l, bn, wq, aq, relu=False, res=None, extra=None, w_override=None):
        w, b = _fold(l, bn) if w_override is None else w_override
        if name in wq or "all" in wq:
            w = q(w)
        y = x @ w.double().t() + b.double()
        if extra is not None:
            y = y + extra
        if res is not None:
            y = y + res
        if relu:
            y = torch.relu(y)
        return q(y.float()).double() if (name in aq or "all" in aq) else y

    def recurrent(wq, aq, in_q):
        x0 = torch.from_numpy(fx["init_hidden"])
        x = (q(x0) if in_q else x0).double()
        act = torch.from_numpy(fx["action"]).reshape(-1)
        w1, b1 = _fold(dyn.fc1, dyn.bn1)
        y = lin("d1", x, None, None, wq, aq, True, extra=w1[:, H:].t()[act].double(), w_override=(w1[:, :H], b1))
        y = lin("d2", y, dyn.fc2, dyn.bn2, wq, aq, True)
        s = lin("d3", y, dyn.fc3, dyn.bn3, wq, aq, True, res=x)
        r = lin("r3", lin("r2", lin("r1", s, rw[0], rw[1], wq, aq, True), rw[3], rw[4], wq, aq, True), rw[6], None, wq, aq)
        v = lin("v3", lin("v2", lin("v1", s, va[0], va[1], wq, aq, True), va[3], va[4], wq, aq, True), va[6], None, wq, aq)
        rv = inverse_scalar_transform(r.float(), -sup, sup).reshape(-1).numpy()
        vv = inverse_scalar_transform(v.float(), -sup, sup).reshape(-1).numpy()
        return "value %s  reward %s  hidden %s" % (err(vv, fx["rec_value"]), err(rv, fx["rec_reward"]), err(s.numpy(), fx["rec_hidden"]))

    def initial_value(wq, aq):
        obs = torch.from_numpy(np.unpackbits(ac["wide_obs_bits"], axis=1)[:, :int(fx["D"]) * int(fx["stack"])].astype(np.float32)).double()
        x = lin("p0", obs, rep[0], rep[1], wq, aq, True)
        y = lin("p1", x, rep[3].fc1, rep[3].bn1, wq, aq, True)
        x = lin("p2", y, rep[3].fc2, rep[3].bn2, wq, aq, True, res=x)
        x = lin("p3", x, rep[4], rep[5], wq, aq, True)
        y = lin("p4", x, rep[7].fc1, rep[7].bn1, wq, aq, True)
        s = lin("p5", y, rep[7].fc2, rep[7].bn2, wq, aq, True, res=x)
        v = lin("v3", lin("v2", lin("v1", s, va[0], va[1], wq, aq, True), va[3], va[4], wq, aq, True), va[6], None, wq, aq)
        return err(inverse_scalar_transform(v.float(), -sup, sup).reshape(-1).numpy(), ac["wide_fp32_init_value"])

    print("errors are max/mean/rms of |got - ref| / max(1, |ref|); format %s" % args.dtype)
    print("== recurrent inference, 32 golden rows (fp32 truth)")
    rec_layers = ["d1", "d2", "d3", "r1", "r2", "r3", "v1", "v2", "v3"]
    print("nothing rounded                  ", recurrent(set(), set(), False))
    print("input hidden state rounded only  ", recurrent(set(), set(), True))
    print("as the kernel (logits fp32)      ", recurrent({"all"}, set(rec_layers) - {"r3", "v3"}, True))
    print("as the kernel before r03 (logits rounded)", recurrent({"all"}, {"all"}, True))
    for L in rec_layers:
        print("weights of %s only               " % L, recurrent({L}, set(), False))
    for L in rec_layers:
        print("output activations of %s only    " % L, recurrent(set(), {L}, False))
    print("== root value, wide sample (256 rows, fp32 truth)")
    init_layers = ["p0", "p1", "p2", "p3", "p4", "p5", "v1", "v2", "v3"]
    print("as the kernel (logits fp32)      ", initial_value({"all"}, set(init_layers) - {"v3"}))
    print("the reference under fp16 autocast", err(ac["wide_autocast_init_value"], ac["wide_fp32_init_value"]))
    for L in init_layers:
        print("weights of %s only  %s | output activations of %s only  %s" % (L, initial_value({L}, set()), L, initial_value(set(), {L})))


if __name__ == "__main__":
    main()
