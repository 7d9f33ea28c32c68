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


def golden_net_error(game, dtype, device="cuda", fused=None):
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
    eng = InferenceEngine(net, sup, dtype=dtype, device=device, fused=fused)
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
    # the same outputs against the reference run under fp16 autocast -- the precision it searches with (core/mcts.py:38-40;
    # tests/golden/nets_<game>_autocast.npz, tools/gen_golden.py::gen_nets_autocast) -- and, as the yardstick, the
    # reference-under-autocast's own distance from its fp32 outputs
    ac = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nets_%s_autocast.npz" % game)))
    got = dict(init_value=v0, init_logits=l0, init_hidden=h0, rec_value=v1, rec_reward=r1, rec_logits=l1, rec_hidden=h1)
    vs_ac, ref_ac = {}, {}
    for name, g_ in got.items():
        g = g_.float().cpu().numpy().astype(np.float64).reshape(-1)
        a = ac[name].astype(np.float64).reshape(-1)
        w = np.asarray(fx[name], np.float64).reshape(-1)
        e1, e2 = np.abs(g - a) / np.maximum(1.0, np.abs(a)), np.abs(a - w) / np.maximum(1.0, np.abs(w))
        vs_ac[name] = {"max": float(e1.max()), "mean": float(e1.mean())}
        ref_ac[name] = {"max": float(e2.max()), "mean": float(e2.mean())}
    out["vs_reference_autocast"] = {"fields": vs_ac, "worst": max(v["max"] for v in vs_ac.values())}
    out["reference_autocast_vs_fp32"] = {"fields": ref_ac, "worst": max(v["max"] for v in ref_ac.values())}
    # ... and the same two comparisons on the fixture's wide sample (256 rows; recurrent inputs exact in fp16): statistics
    out["wide"] = _wide_errors(eng, ac, D * stack, device)
    out["fused"] = eng.fused is not None
    return out


def _wide_errors(eng, ac, obs_width, device):
    import torch
    obs = torch.from_numpy(np.unpackbits(ac["wide_obs_bits"], axis=1)[:, :obs_width].astype(np.float32)).to(device)
    hid = torch.from_numpy(ac["wide_hidden_in"]).to(device).to(eng.dtype)
    act = torch.from_numpy(ac["wide_action"]).reshape(-1).to(device)
    N = hid.shape[0]
    v0, l0, _ = eng.initial(obs)
    if eng.fused is not None:
        h1 = torch.zeros(N, eng.H, dtype=eng.dtype, device=device)
        r1, v1 = torch.zeros(N, device=device), torch.zeros(N, device=device)
        l1 = torch.zeros(N, eng.A, device=device)
        eng.fused(hid, None, act.to(torch.int32), h1, r1, v1, l1)
    else:
        v1, r1, l1, _ = eng.recurrent(hid, act)
    got = dict(init_value=v0, init_logits=l0, rec_value=v1, rec_reward=r1, rec_logits=l1)

    def stats(g, w):
        g, w = np.asarray(g, np.float64).reshape(-1), np.asarray(w, np.float64).reshape(-1)
        e = np.abs(g - w) / np.maximum(1.0, np.abs(w))
        return {"max": float(e.max()), "mean": float(e.mean()), "rms": float(np.sqrt((e * e).mean()))}
    res = {"rows": int(N), "got_vs_fp32": {}, "reference_autocast_vs_fp32": {}}
    for k, g in got.items():
        res["got_vs_fp32"][k] = stats(g.float().cpu().numpy(), ac["wide_fp32_" + k])
        res["reference_autocast_vs_fp32"][k] = stats(ac["wide_autocast_" + k], ac["wide_fp32_" + k])
    return res


def search_fixture(game):
    """tests/golden/search_<game>_autocast.npz (tools/gen_golden.py::gen_search_autocast): 512 roots searched by the REFERENCE's
    nets + tree in fp32 and under fp16 autocast."""
    import os
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "search_%s_autocast.npz" % game)))


def _divergence(d0, v0, d1, v1, sims):
    d0, d1 = np.asarray(d0, np.float64), np.asarray(d1, np.float64)
    tv = 0.5 * np.abs(d0 - d1).sum(1) / sims
    return {"argmax_agreement": float((d0.argmax(1) == d1.argmax(1)).mean()), "visit_tv_mean": float(tv.mean()),
            "visit_tv_max": float(tv.max()), "identical_visit_counts": float((tv == 0).mean()),
            "root_value_abs_diff_mean": float(np.abs(np.asarray(v0, np.float64) - np.asarray(v1, np.float64)).mean())}


def search_divergence(game, dtype, roots=512, simulations=50, device="cuda", seed=0):
    """What the inference format does to the quantity the search produces: the same `roots` root positions (random binary
    observation windows, the golden weight recipe) searched `simulations - 1` times with the fp32 engine (launch-per-phase
    search, hipBLASLt GEMMs: inside north_star's 1e-3 of the reference nets) and with the engine bench.py times in `dtype`
    (the persistent kernel with the fused MFMA inference), same Dirichlet noise, same tie-break seed.
    Returns the share of roots whose most-visited action agrees, the mean / max total-variation distance between the two
    visit distributions, and the mean |root value difference|.
    With the default root set (512 roots, seed 0, 50 simulations: the one tools/gen_golden.py::gen_search_autocast searched with
    the reference's own nets and tree) the result also carries, under "reference", the same figures for the REFERENCE's fp16-autocast
    search against its fp32 search -- the yardstick -- and each engine's search against the reference's search of its precision."""
    import torch
    from hanabizero_amd import cytree
    from hanabizero_amd.config import make_config
    from hanabizero_amd.mcts import MCTS
    from hanabizero_amd.model import InferenceEngine
    stack = 1 if game == "Hanabi-Small" else 4
    cfg = make_config(game, simulations=simulations, stack=stack, p_mcts_num=roots)
    net = cfg.get_uniform_network()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
    net.eval()
    rng = np.random.RandomState(seed)
    A = cfg.action_space_size
    obs_h = (rng.rand(roots, cfg.obs_shape) < 0.3)
    noise_h = rng.dirichlet([cfg.root_dirichlet_alpha] * A, roots).astype(np.float32)
    legal_h = (rng.rand(roots, A) < 0.7).astype(np.uint8)
    legal_h[:, 0] = 1
    fx = search_fixture(game) if (roots, seed, simulations) == (512, 0, 50) else None
    if fx is not None:  # (the fixture's inputs ARE this recipe's: checked, then used)
        assert np.array_equal(np.unpackbits(fx["obs_bits"], axis=1)[:, :cfg.obs_shape].astype(bool), obs_h)
        assert np.array_equal(fx["noise"], noise_h) and np.array_equal(fx["legal"], legal_h) and int(fx["tie_seed"]) == seed + 1
    obs = torch.from_numpy(obs_h.astype(np.float32)).to(device)
    noise = torch.from_numpy(noise_h).to(device)
    legal = torch.from_numpy(legal_h).to(device)
    res = {}
    for name, dt in (("ref", torch.float32), ("got", dtype)):
        eng = InferenceEngine(net, cfg.value_support.max, dtype=dt, device=device)
        _, logits0, hidden0 = eng.initial(obs)
        r = cytree.Roots(roots, A, simulations, tie_seed=seed + 1)
        r.prepare(cfg.root_exploration_fraction, noise, torch.zeros(roots, device=device), logits0.float(), legal)
        MCTS(cfg).run_multi(r, eng, hidden0)
        res[name] = (r.distributions_tensor().cpu().numpy().astype(np.float64), r.values_tensor().cpu().numpy().astype(np.float64))
    (d0, v0), (d1, v1) = res["ref"], res["got"]
    assert (d0.sum(1) == simulations - 1).all() and (d1.sum(1) == simulations - 1).all()
    out = {"roots": roots, "simulations": simulations - 1}
    out.update(_divergence(d0, v0, d1, v1, simulations - 1))
    if fx is not None:
        out["reference"] = {
            # the reference's own fp16-autocast search against its fp32 search (core/mcts.py:38-40): the yardstick
            "autocast_vs_fp32": _divergence(fx["dist_fp32"], fx["values_fp32"], fx["dist_autocast"], fx["values_autocast"], simulations - 1),
            # the fp32 engine's search against the reference's fp32 search (nets 7e-5 apart: a near-tie flips now and then)
            "fp32_engine_vs_reference_fp32": _divergence(fx["dist_fp32"], fx["values_fp32"], d0, v0, simulations - 1),
            # the benched engine's search against the reference's search at each precision
            "engine_vs_reference_fp32": _divergence(fx["dist_fp32"], fx["values_fp32"], d1, v1, simulations - 1),
            "engine_vs_reference_autocast": _divergence(fx["dist_autocast"], fx["values_autocast"], d1, v1, simulations - 1)}
    return out
