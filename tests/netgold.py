"""tests/netgold.py -- deterministic weight recipe shared by tools/gen_golden.py and tests/test_model.py.

The nets fixtures hold only inputs and the reference's outputs; the (large) state_dict is regenerated from this
recipe on both sides, keyed by parameter name, so it does not have to be committed.
"""
import zlib

import numpy as np


def make_weight(key, shape):
    rng = np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, np.int64)
    if key.endswith("running_var"):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith("running_mean"):
        return rng.normal(0, 0.2, shape).astype(np.float32)
    if key.endswith("bias"):
        return rng.normal(0, 0.1, shape).astype(np.float32)
    if len(shape) == 2:  # Linear.weight [out, in]
        return rng.normal(0, 1.0 / np.sqrt(shape[1]), shape).astype(np.float32)
    return rng.uniform(0.8, 1.2, shape).astype(np.float32)  # BatchNorm1d.weight


def fill_state_dict(state_dict):
    """Returns {key: numpy array} for every entry of a torch state_dict (shapes taken from it)."""
    return {k: make_weight(k, tuple(v.shape)) for k, v in state_dict.items()}


def golden_net_error(game, dtype, device="cuda"):
    """Error of the product's inference path in `dtype` -- InferenceEngine.initial (the fused tail of the root inference
    where it exists) and the fused MFMA recurrent kernel the search launches -- against the reference nets' own fp32
    outputs (tests/golden/nets_<game>.npz, written by tools/gen_golden.py from config/hanabi_control/model.py).
    Returns {output: {"max": worst |got - ref| / max(1, |ref|), "mean": the mean of it}} plus "worst" = the largest max.
    Used by tests/test_model.py (asserted per dtype) and by bench.py (printed as `net_error` beside the throughput)."""
    import os

    import torch
    from hanabizero_amd.model import InferenceEngine, MuZeroNet, MuZeroNetFull, inverse_scalar_transform
    fx = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nets_%s.npz" % game)))
    D, A, sup, stack = int(fx["D"]), int(fx["A"]), int(fx["support"]), int(fx["stack"])
    inv = lambda x: inverse_scalar_transform(x, -sup, sup)
    net = (MuZeroNet if game == "Hanabi-Small" else MuZeroNetFull)(D * stack, A, 2 * sup + 1, 2 * sup + 1, inv, inv)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
    net.eval()
    eng = InferenceEngine(net, sup, dtype=dtype, device=device)
    obs = torch.from_numpy(fx["obs"]).to(device)
    hid = torch.from_numpy(fx["init_hidden"]).to(device).to(dtype)
    act = torch.from_numpy(fx["action"]).reshape(-1).to(device)
    N = hid.shape[0]
    v0, l0, h0 = eng.initial(obs)
    if eng.fused is not None:
        h1 = torch.zeros(N, eng.H, dtype=dtype, device=device)
        r1, v1 = torch.zeros(N, device=device), torch.zeros(N, device=device)
        l1 = torch.zeros(N, eng.A, device=device)
        eng.fused(hid, None, act.to(torch.int32), h1, r1, v1, l1)
    else:
        v1, r1, l1, h1 = eng.recurrent(hid, act)
    out = {}
    for name, got, want in [("init_value", v0, fx["init_value"]), ("init_logits", l0, fx["init_logits"]),
                            ("init_hidden", h0, fx["init_hidden"]), ("rec_value", v1, fx["rec_value"]),
                            ("rec_reward", r1, fx["rec_reward"]), ("rec_logits", l1, fx["rec_logits"]),
                            ("rec_hidden", h1, fx["rec_hidden"])]:
        g = got.float().cpu().numpy().astype(np.float64).reshape(-1)
        w = np.asarray(want, np.float64).reshape(-1)
        e = np.abs(g - w) / np.maximum(1.0, np.abs(w))
        out[name] = {"max": float(e.max()), "mean": float(e.mean())}
    out["worst"] = max(v["max"] for v in out.values())
    out["fused"] = eng.fused is not None
    return out
