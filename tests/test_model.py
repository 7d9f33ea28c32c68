"""Nets: hanabizero_amd.model against the reference's own networks (tests/golden/nets_*.npz were produced by
config/hanabi_control/model.py loaded by file path, CPU fp32, eval mode; weights from tests/netgold.py)."""
import os

import numpy as np
import pytest
import torch

from tests.netgold import fill_state_dict

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build(game):
    from hanabizero_amd.model import MuZeroNet, MuZeroNetFull, inverse_scalar_transform
    fx = dict(np.load(os.path.join(GOLD, "nets_%s.npz" % game)))
    D, A, sup, stack = int(fx["D"]), int(fx["A"]), int(fx["support"]), int(fx["stack"])
    inv = lambda x: inverse_scalar_transform(x, -sup, sup)
    cls = MuZeroNet if game == "Hanabi-Small" else MuZeroNetFull
    net = cls(D * stack, A, 2 * sup + 1, 2 * sup + 1, inv, inv)
    assert list(net.state_dict().keys()) == [str(k) for k in fx["sd_keys"]], "state_dict keys differ from the reference"
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
    net.eval()
    return net, fx, sup


def _close(a, b, tol):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) <= tol


@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
def test_module_matches_reference_fp32_cpu(game):
    net, fx, _ = build(game)
    with torch.no_grad():
        o0 = net.initial_inference(torch.from_numpy(fx["obs"]))
        o1 = net.recurrent_inference(torch.from_numpy(o0.hidden_state), torch.from_numpy(fx["action"]))
    for got, want in [(o0.value, fx["init_value"]), (o0.policy_logits, fx["init_logits"]), (o0.hidden_state, fx["init_hidden"]),
                      (o1.value, fx["rec_value"]), (o1.reward, fx["rec_reward"]), (o1.policy_logits, fx["rec_logits"]),
                      (o1.hidden_state, fx["rec_hidden"])]:
        assert _close(got, want, 1e-5)
    assert o0.reward == [0.0] * fx["obs"].shape[0]  # core/model.py:71


@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
def test_inference_engine_fp32_cpu_within_1e3(game):
    """BatchNorm folding, action-column lookup, the fused head GEMM and the GEMM operand layout change rounding only:
    1e-3 is the north-star tolerance for net outputs; fp32 logits and hidden states land orders of magnitude inside it
    (1e-4 asserted, ~2e-6 measured), the value / reward scalars -- softmax . support through h^-1, which amplifies the
    logits' rounding -- at ~1e-4 (1e-3 asserted)."""
    from hanabizero_amd.model import InferenceEngine
    net, fx, sup = build(game)
    eng = InferenceEngine(net, sup, dtype=torch.float32, device="cpu")
    v0, l0, h0 = eng.initial(torch.from_numpy(fx["obs"]))
    assert _close(v0, fx["init_value"], 1e-3) and _close(l0, fx["init_logits"], 1e-4) and _close(h0, fx["init_hidden"], 1e-4)
    v1, r1, l1, h1 = eng.recurrent(torch.from_numpy(fx["init_hidden"]), torch.from_numpy(fx["action"]).reshape(-1))
    assert _close(v1, fx["rec_value"], 1e-3) and _close(r1, fx["rec_reward"], 1e-3)
    assert _close(l1, fx["rec_logits"], 1e-4) and _close(h1, fx["rec_hidden"], 1e-4)
    pool_slot = torch.zeros_like(h1)
    eng.recurrent(torch.from_numpy(fx["init_hidden"]), torch.from_numpy(fx["action"]).reshape(-1), hidden_out=pool_slot)
    assert torch.equal(pool_slot, h1)


@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
def test_fused_recurrent_heads_equals_unfused_cpu(game):
    """[state | one-hot | 0] rows through the padded first layer + batched head tails == the layer-by-layer path."""
    from hanabizero_amd.model import InferenceEngine, inverse_scalar_transform
    net, fx, sup = build(game)
    eng = InferenceEngine(net, sup, dtype=torch.float32, device="cpu")
    hid, act = torch.from_numpy(fx["init_hidden"]), torch.from_numpy(fx["action"]).reshape(-1)
    N = hid.shape[0]
    net_in = torch.zeros(N, eng.H + eng.onehot_cols)
    net_in[:, :eng.H] = hid
    net_in[torch.arange(N), eng.H + act] = 1
    out = torch.empty(N, eng.H)
    r_log, v_log, p_log = eng.recurrent_heads(net_in, out)
    assert _close(out, fx["rec_hidden"], 1e-4) and _close(p_log[:, :eng.A], fx["rec_logits"], 1e-4)
    assert _close(inverse_scalar_transform(v_log[:, :eng.V], -sup, sup), fx["rec_value"], 1e-4)
    assert _close(inverse_scalar_transform(r_log[:, :eng.V], -sup, sup), fx["rec_reward"], 1e-4)


def test_zero_initialised_heads_like_the_reference():
    from hanabizero_amd.model import MuZeroNetFull
    net = MuZeroNetFull(785 * 4, 20, 201, 201, None, None)
    for head in (net._prediction_value, net._dynamics_reward, net._prediction_actor):
        assert float(head[-1].weight.abs().sum()) == 0.0 and float(head[-1].bias.abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
def test_inference_engine_on_gpu(game):
    """fp32 on the GPU stays within 1e-3 of the reference's fp32 outputs (relative to max(1,|ref|)).
    bf16 / fp16 engines (8 / 11 significand bits) are checked at the accuracy those formats allow; the reference
    itself runs these nets under fp16 autocast (core/mcts.py:38-40)."""
    from hanabizero_amd.model import InferenceEngine
    net, fx, sup = build(game)
    obs = torch.from_numpy(fx["obs"]).cuda()
    hid = torch.from_numpy(fx["init_hidden"]).cuda()
    act = torch.from_numpy(fx["action"]).reshape(-1).cuda()
    for dtype, tol in [(torch.float32, 1e-3), (torch.float16, 1e-2), (torch.bfloat16, 6e-2)]:
        eng = InferenceEngine(net, sup, dtype=dtype, device="cuda")
        v0, l0, h0 = eng.initial(obs)
        v1, r1, l1, h1 = eng.recurrent(hid.to(dtype), act)
        # the search loop's fused form: HIP residual/ReLU glue, batched head tails, HIP scalar transform
        N = hid.shape[0]
        net_in = torch.zeros(N, eng.H + eng.onehot_cols, dtype=dtype, device="cuda")
        net_in[:, :eng.H] = hid.to(dtype)
        net_in[torch.arange(N), eng.H + act] = 1
        h2 = torch.empty(N, eng.H, dtype=dtype, device="cuda")
        r_log, v_log, p_log = eng.recurrent_heads(net_in, h2)
        v2, r2 = eng.support_to_scalar(v_log), eng.support_to_scalar(r_log)
        for got, want in [(v0, fx["init_value"]), (l0, fx["init_logits"]), (h0, fx["init_hidden"]), (v1, fx["rec_value"]),
                          (r1, fx["rec_reward"]), (l1, fx["rec_logits"]), (h1, fx["rec_hidden"]),
                          (v2, fx["rec_value"]), (r2, fx["rec_reward"]), (p_log[:, :eng.A], fx["rec_logits"]), (h2, fx["rec_hidden"])]:
            assert _close(got.float().cpu().numpy(), want, tol), (dtype, float(np.max(np.abs(got.float().cpu().numpy().reshape(-1) - np.asarray(want).reshape(-1)))))


# Measured worst / mean errors of the product path against the reference's fp32 outputs on the golden inputs
# (tests/netgold.py::golden_net_error; relative to max(1, |ref|)) and the bounds asserted: <= 1.5x the measured worst field
# (r03, head logits fp32 through the scalar transform: fp32 7.2e-5 / 3.7e-6, fp16 5.5e-3 / 2.0e-3, bf16 3.9e-2 / 1.4e-2).
# fp32 meets north_star's 1e-3.  fp16 is the reference's own search format (autocast, core/mcts.py:38-40): its bound is DERIVED
# FROM THE REFERENCE -- the fp16 engine must be no further from the reference's fp32 outputs than the reference itself is when
# run under fp16 autocast (tests/golden/nets_*_autocast.npz: worst 4.4e-3 Small / 6.5e-3 Full) -- and 1e-3 is out of reach of
# any engine whose hidden-state pool is 16-bit: rounding the golden input state to fp16 alone moves the reward scalar by 2.9e-3
# (tools/net_error_ablation.py).  bench.py prints the same measurement as `net_error` for the dtype it ran.
NET_ERROR_BOUND = {
    # dtype: (bound on every output's worst element, bound on every output's mean)
    torch.float32: (1e-3, 1e-4),
    torch.float16: (8e-3, 3e-3),
    torch.bfloat16: (5.5e-2, 2e-2),
}
# search level (tests/netgold.py::search_divergence, 512 roots x 49 simulations against the fp32 engine's search):
# (minimum share of roots with the same most-visited action, maximum mean total-variation distance of the visit distributions)
# fp16 -- the reference's own search precision -- is held to a yardstick DERIVED FROM THE REFERENCE (r04): the same 512 roots searched
# by the reference's nets + tree in fp32 and under fp16 autocast (tests/golden/search_*_autocast.npz, tools/gen_golden.py::
# gen_search_autocast: Small 0.975 / 0.0027, Full 0.951 / 0.0094).  bf16 is not a format the reference searches in: its bound stays
# the measured one (r03 over several root sets: 0.857-0.927 / 0.0097-0.029).
SEARCH_DIVERGENCE_BOUND = {torch.bfloat16: (0.78, 0.05)}
# one binomial standard deviation of an agreement share near 0.95 measured on 512 roots: the two searches being compared are
# different computations of the same roots, each flipping its own near-ties
SEARCH_AGREEMENT_SLACK, SEARCH_TV_SLACK = 0.01, 1.1


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_benched_inference_path_error_against_reference_goldens(game, dtype):
    """The path bench.py times in each --dtype (fused MFMA kernels for bf16 / fp16, GEMM chain for fp32) against the
    reference nets' fp32 outputs: the measured error is printed and bounded per format; the fp16 engine -- the reference's
    search format -- additionally against the reference's own fp16-autocast outputs."""
    from tests.netgold import golden_net_error
    err = golden_net_error(game, dtype)
    assert err["fused"] == (dtype != torch.float32)
    mx, mean = NET_ERROR_BOUND[dtype]
    fields = {k: v for k, v in err.items() if isinstance(v, dict) and "max" in v}
    print("net error %s %s: %s; vs reference under fp16 autocast %.3g; reference autocast vs its fp32 %.3g" % (
        game, dtype, fields, err["vs_reference_autocast"]["worst"], err["reference_autocast_vs_fp32"]["worst"]))
    assert len(fields) == 7
    for k, v in fields.items():
        assert v["max"] <= mx and v["mean"] <= mean, (game, dtype, k, v)
    if dtype == torch.float16:
        ref = err["reference_autocast_vs_fp32"]
        # no further from fp32 than the reference's own search-time outputs are (worst element over all outputs) ...
        assert err["worst"] <= ref["worst"], (err["worst"], ref["worst"])
        # ... and on the wide sample (256 rows: the 32-row means of the amplified scalars are noise) the engine's rms error, as
        # a ratio to the reference-under-autocast's own rms error of the same output, is below 1 on average over the outputs and
        # nowhere above 1.3 (measured r03: Small 0.71 - 1.05, Full 0.76 - 1.24; the outliers are the value scalars, where the
        # FIXED rounding errors of a head's weights act on positive-mean ReLU inputs as a near-constant shift -- which way it
        # falls is a property of the weight set, tools/net_error_ablation.py)
        w = err["wide"]
        ratios = {k: v["rms"] / w["reference_autocast_vs_fp32"][k]["rms"] for k, v in w["got_vs_fp32"].items()}
        print("wide sample, rms(engine - fp32) / rms(reference autocast - fp32):", ratios)
        assert max(ratios.values()) <= 1.3 and sum(ratios.values()) / len(ratios) <= 1.0, ratios
        # the two 16-bit computations of the same nets differ from each other by no more than each differs from fp32
        assert err["vs_reference_autocast"]["worst"] <= 1.5 * ref["worst"]
    if dtype == torch.float32:
        assert abs(err["vs_reference_autocast"]["worst"] - err["reference_autocast_vs_fp32"]["worst"]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
def test_fp16_pair_inference_is_inside_the_contract_tolerance(game):
    """north_star's figure for the nets is 1e-3 against the reference's fp32 outputs.  The engine that meets it with a hand-written
    kernel: fp32 pool and root inference, recurrent inference = the MFMA kernel's fp16-pair build (include/hz_mlp.h, HZ_F16X2).
    Tolerance 1e-3 on max |got - ref| / max(1, |ref|) over every output of tests/golden/nets_<game>.npz (measured: 7e-5 Full,
    1.6e-4 Small -- the value scalar; hidden states and logits 2e-6), and no worse than 2 x the library-GEMM fp32 engine + 2e-4."""
    from tests.netgold import golden_net_error
    err = golden_net_error(game, torch.float32, fused="fp16x2")
    base = golden_net_error(game, torch.float32)
    assert err["fused"] and not base["fused"]
    fields = {k: v["max"] for k, v in err.items() if isinstance(v, dict) and "max" in v}
    print("net error %s fp16 pairs: %s (fp32 GEMM chain: worst %.3g)" % (game, fields, base["worst"]))
    assert len(fields) == 7 and err["worst"] <= 1e-3, fields
    assert err["worst"] <= 2 * base["worst"] + 2e-4
    for k in ("rec_hidden", "rec_logits"):   # (no scalar transform behind these: the kernel's own arithmetic)
        assert fields[k] <= 1e-5, (k, fields[k])


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_search_level_divergence_of_the_16_bit_engines(game, dtype):
    """What the net error does to the quantity the contract cares about: visit counts and chosen actions of the benched engine's
    search (persistent kernel, fused MFMA inference) against the fp32 engine's search of the same roots."""
    from tests.netgold import search_divergence
    d = search_divergence(game, dtype, roots=512)
    print("search divergence %s %s: %s" % (game, dtype, d))
    ref = d["reference"]
    # the fp32 engine's search IS the reference's fp32 search up to the rare near-tie that nets 7e-5 apart flip
    assert ref["fp32_engine_vs_reference_fp32"]["argmax_agreement"] >= 0.99 and ref["fp32_engine_vs_reference_fp32"]["visit_tv_mean"] <= 0.002, ref
    if dtype == torch.float16:
        # no further from the fp32 search than the reference's own fp16-autocast search is from its fp32 search ...
        y = ref["autocast_vs_fp32"]
        assert d["argmax_agreement"] >= y["argmax_agreement"] - SEARCH_AGREEMENT_SLACK, (d, y)
        assert d["visit_tv_mean"] <= y["visit_tv_mean"] * SEARCH_TV_SLACK, (d, y)
        # ... and measured against the REFERENCE's fp32 search itself (not the fp32 engine's) no worse either
        e = ref["engine_vs_reference_fp32"]
        assert e["argmax_agreement"] >= y["argmax_agreement"] - SEARCH_AGREEMENT_SLACK and e["visit_tv_mean"] <= y["visit_tv_mean"] * SEARCH_TV_SLACK, (e, y)
    else:
        agree, tv = SEARCH_DIVERGENCE_BOUND[dtype]
        assert d["argmax_agreement"] >= agree and d["visit_tv_mean"] <= tv, d


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
@pytest.mark.parametrize("N", [32, 100, 4096, 8192])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_mfma_recurrent_kernel(game, N, dtype):
    """hz_mlp_recurrent (one hand-written MFMA kernel) against the layer-by-layer PyTorch path of the same format it
    replaces (same rounding points: a few ulps apart) and, on the golden inputs, against the reference's fp32 outputs."""
    from hanabizero_amd.model import FusedRecurrent, InferenceEngine
    net, fx, sup = build(game)
    eng = InferenceEngine(net, sup, dtype=dtype, device="cuda")
    fused = FusedRecurrent(net, eng)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    g = torch.Generator(device="cuda").manual_seed(N)
    B = fx["init_hidden"].shape[0]
    hid = torch.from_numpy(fx["init_hidden"]).cuda().to(dtype)
    act = torch.from_numpy(fx["action"]).reshape(-1).cuda()
    if N > B:
        hid = torch.cat([hid, (torch.rand(N - B, eng.H, device="cuda", generator=g) * 2).to(dtype)])
        act = torch.cat([act, torch.randint(0, eng.A, (N - B,), device="cuda", generator=g)])
    net_in = torch.zeros(N, eng.H + eng.onehot_cols, dtype=dtype, device="cuda")
    net_in[:, :eng.H] = hid
    net_in[torch.arange(N), eng.H + act] = 1
    h_ref = torch.empty(N, eng.H, dtype=dtype, device="cuda")
    r_log, v_log, p_log = eng.recurrent_heads(net_in, h_ref)
    r_ref, v_ref = eng.support_to_scalar(r_log), eng.support_to_scalar(v_log)
    h = torch.zeros(N, eng.H, dtype=dtype, device="cuda")
    r, v = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    p = torch.empty(N, eng.A, device="cuda")
    fused(hid, None, act.to(torch.int32), h, r, v, p)
    torch.cuda.synchronize()
    # ... and through the pool gather: pool[ix[i], i] = hid[i]
    S = 5
    ixs = torch.randint(0, S, (N,), device="cuda", generator=g).to(torch.int32)
    pool = (torch.rand(S, N, eng.H, device="cuda", generator=g)).to(dtype)
    pool[ixs.long(), torch.arange(N, device="cuda")] = hid
    h2, r2, v2, p2 = torch.zeros_like(h), torch.zeros_like(r), torch.zeros_like(v), torch.zeros_like(p)
    fused(pool, ixs, act.to(torch.int32), h2, r2, v2, p2)
    assert torch.equal(h2, h) and torch.equal(r2, r) and torch.equal(v2, v) and torch.equal(p2, p)

    def err(a, b):
        a, b = a.float(), b.float()
        return float(((a - b).abs() / b.abs().clamp(min=1.0)).max())
    # fp32 truth on the same inputs (fp32 engine, torch scalar transform)
    from hanabizero_amd.model import inverse_scalar_transform
    e32 = InferenceEngine(net, sup, dtype=torch.float32, device="cuda")
    h32 = torch.empty(N, eng.H, dtype=torch.float32, device="cuda")
    r32l, v32l, p32 = e32.recurrent_heads(net_in.float(), h32)
    r32 = inverse_scalar_transform(r32l[:, :eng.V], -sup, sup).reshape(-1)
    v32 = inverse_scalar_transform(v32l[:, :eng.V], -sup, sup).reshape(-1)
    for name, got, torch_same, truth in [("hidden", h, h_ref, h32), ("policy", p, p_log[:, :eng.A], p32[:, :eng.A]),
                                         ("reward", r, r_ref, r32), ("value", v, v_ref, v32)]:
        e_fused, e_torch = err(got, truth), err(torch_same, truth)
        m_fused = float(((got.float() - truth).abs() / truth.abs().clamp(min=1.0)).mean())
        m_torch = float(((torch_same.float() - truth).abs() / truth.abs().clamp(min=1.0)).mean())
        # the hand-written kernel is as close to fp32 as the bf16 PyTorch path it replaces (same rounding points):
        # mean error within 1.3x, worst element within 2.5x (maxima of a few thousand bf16 roundings are noisy)
        assert m_fused <= 1.3 * m_torch + 1e-4, (name, m_fused, m_torch)
        if N <= B:  # in-distribution (golden) inputs: worst element too (maxima over out-of-distribution rows are noise)
            assert e_fused <= max(2.5 * e_torch, 5 * ulp), (name, e_fused, e_torch)
            assert e_fused < 20 * ulp, (name, e_fused)
    # mean error is far below the worst-case ulp bound: no systematic (indexing) error
    assert float((h.float() - h32).abs().mean()) < 1.5 * ulp
    mx = NET_ERROR_BOUND[dtype][0]
    assert err(h[:B], torch.from_numpy(fx["rec_hidden"]).cuda()) < mx
    assert err(v[:B], torch.from_numpy(fx["rec_value"]).reshape(-1).cuda()) < mx
    assert err(p[:B], torch.from_numpy(fx["rec_logits"]).cuda()) < mx


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
@pytest.mark.parametrize("shape", [(8, 4), (16, 2)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_mfma_kernel_workgroup_shapes_give_identical_bits(game, shape, dtype):
    """The same layer chain cut for another workgroup shape (16 waves x 2 tiles: the persistent search kernel) sums
    every output column over k in the same order, so nothing may differ from the 4 x 4 kernel."""
    from hanabizero_amd.model import InferenceEngine
    net, fx, sup = build(game)
    eng = InferenceEngine(net, sup, dtype=dtype, device="cuda")
    N, S = 1000, 4
    g = torch.Generator(device="cuda").manual_seed(5)
    pool = (torch.rand(S, N, eng.H, device="cuda", generator=g) * 2).to(dtype)
    ix = torch.randint(0, S, (N,), device="cuda", generator=g).to(torch.int32)
    act = torch.randint(0, eng.A, (N,), device="cuda", generator=g).to(torch.int32)
    outs = []
    for sh in ((4, 4), shape):
        f = eng.fused_shape(*sh)
        h = torch.zeros(N, eng.H, dtype=dtype, device="cuda")
        r, v, p = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, eng.A, device="cuda")
        f(pool, ix, act, h, r, v, p)
        outs.append((h, r, v, p))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _job_table(chain):
    import ctypes
    from hanabizero_amd._lib import MlpJob
    raw = (chain._host["jobs"] if chain._host is not None else chain.jobs.cpu()).numpy().tobytes()
    return (MlpJob * (len(raw) // ctypes.sizeof(MlpJob))).from_buffer_copy(raw)


@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
def test_job_table_synchronisation_is_proven_and_the_proof_notices_a_missing_wait(game):
    """hanabizero_amd/mlp_sync.py: the 16 x 2 chain of the recurrent inference ends up with no workgroup barrier between its
    passes (blockwise boundaries + per-job waits); the checker accepts the table as built, and rejects it as soon as one
    token is taken away or a blockwise pass is declared to need nothing."""
    from hanabizero_amd import mlp_sync
    from hanabizero_amd.model import FusedRecurrent, InferenceEngine, MLP_BARRIER, MLP_BLOCKWISE, MLP_F32_OUT, MLP_STORE_HIDDEN, MLP_WAITS
    net, fx, sup = build(game)
    eng = InferenceEngine(net, sup, dtype=torch.bfloat16, device="cpu", fused=False)
    f = FusedRecurrent(net, eng, 16, 2, host_only=True)
    tab, P = _job_table(f), f.n_jobs
    flags = [tab[p * 16].flags for p in range(P)]
    assert not any(x & MLP_BARRIER for x in flags), flags
    assert sum(1 for x in flags if x & MLP_BLOCKWISE) == 3 and any(x & MLP_WAITS for x in flags)
    cw, H = 32, eng.H
    jobs = []
    for p in range(P):
        hid = [(f.header.hidden_off, f.header.hidden_off + H)] if flags[p] & MLP_STORE_HIDDEN else []
        row = []
        for w in range(16):
            e = tab[p * 16 + w]
            j = mlp_sync.Job(reads=hid, active=False) if e.ks == 0 else mlp_sync.Job(
                reads=[(e.src_off, e.src_off + 32 * e.ks)] + hid + ([] if e.res_off < 0 else [(e.res_off, e.res_off + cw)]),
                writes=[(e.dst_off, e.dst_off + (2 * cw if e.flags & MLP_F32_OUT else cw))])  # (fp32 outputs: 2 image columns each)
            if flags[p] & MLP_WAITS:
                nt = (e.flags >> 8) & 7
                j.tokens = [(((e.producer >> (8 * k)) >> 4) & 15, ((e.producer >> (8 * k)) >> 2) & 3, ((e.producer >> (8 * k)) & 3) + 1)
                            for k in range(nt)]
            row.append(j)
        jobs.append(row)
    pf = [x & (MLP_BARRIER | MLP_BLOCKWISE | MLP_WAITS) for x in flags]
    assert mlp_sync.verify(pf, jobs)
    p, w = next((p, w) for p in range(P) for w in range(16) if jobs[p][w].tokens)
    kept = jobs[p][w].tokens
    jobs[p][w].tokens = kept[1:]
    with pytest.raises(AssertionError, match="may run before"):
        mlp_sync.verify(pf, jobs)
    jobs[p][w].tokens = kept
    q = next(i for i, x in enumerate(pf) if x & MLP_BLOCKWISE)
    with pytest.raises(AssertionError, match="may run before"):
        mlp_sync.verify(pf[:q] + [0] + pf[q + 1:], jobs)


def test_a_wave_that_idles_through_a_blockwise_pass_is_not_ordered_by_it():
    """In the kernel an idle entry of a HZ_MLP_BLOCKWISE pass polls nothing (hz_mlp_dev.h: no barrier there, `continue` before
    any poll), so the proof must not credit it with the pass's producers: a wave that sits a blockwise pass out and then reads
    the producers' columns needs tokens of its own (or a barrier), and a table without them must be rejected.  (Hanabi-Small's
    table has such entries -- waves 12-15 idle in its third blockwise pass; they carry tokens in the pass after.)"""
    from hanabizero_amd import mlp_sync
    J, B, BW, W = mlp_sync.Job, mlp_sync.BARRIER, mlp_sync.BLOCKWISE, mlp_sync.WAITS

    def table():
        p0 = [J(reads=[(0, 512)], writes=[(512 + 32 * w, 544 + 32 * w)]) for w in range(16)]                     # full-width layer
        p1 = [J(reads=[(512, 1024)], writes=[(1024 + 32 * w, 1056 + 32 * w)]) if w < 12 else J(active=False) for w in range(16)]
        p2 = [J(reads=[(512, 1024)], writes=[(1536 + 32 * w, 1568 + 32 * w)]) if w >= 12 else J(active=False) for w in range(16)]
        return [p0, p1, p2]
    jobs = table()
    flags, signal = mlp_sync.plan([0, BW, B], jobs)
    assert (flags[2] & B) or all({(q, g) for (q, g, _c) in jobs[2][w].tokens} == {(0, 0), (0, 1), (0, 2), (0, 3)} for w in range(12, 16)), \
        (flags, [jobs[2][w].tokens for w in range(12, 16)])
    assert (flags[2] & B) or ((flags[2] & W) and 0 in signal)
    # the same table with neither a barrier nor tokens on pass 2: waves 12-15 could read pass 0's columns before they exist
    bare = table()
    with pytest.raises(AssertionError, match="may run before"):
        mlp_sync.verify([0, BW, 0], bare)
    # an ACTIVE entry of the blockwise pass is ordered behind the producers, as before
    ok = table()
    ok[2] = [J(reads=[(1024, 1056)], writes=[(1536, 1568)]) if w == 0 else J(active=False) for w in range(16)]
    assert mlp_sync.verify([0, BW, 0], ok)  # (wave 0 reads its own pass-1 output: program order behind an active blockwise job)


@pytest.mark.gpu
@pytest.mark.parametrize("game", ["Hanabi-Small", "Hanabi-Full"])
@pytest.mark.parametrize("rows", [16, 32])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_blockwise_layer_boundaries_equal_barriers(game, rows, dtype, monkeypatch):
    """16 x 2 chains with HZ_MLP_SIGNAL / HZ_MLP_BLOCKWISE jobs (per-block arrival counters instead of a workgroup barrier
    between full-width layers, include/hz_mlp.h) against the same chain cut with barriers only: the same bits, and the
    job table really carries the flags."""
    import ctypes
    from hanabizero_amd._lib import MlpJob
    from hanabizero_amd.model import FusedRecurrent, InferenceEngine, MLP_BLOCKWISE, MLP_SIGNAL
    net, fx, sup = build(game)
    eng = InferenceEngine(net, sup, dtype=dtype, device="cuda")
    N, S = 1000, 4
    g = torch.Generator(device="cuda").manual_seed(11)
    pool = (torch.rand(S, N, eng.H, device="cuda", generator=g) * 2).to(dtype)
    ix = torch.randint(0, S, (N,), device="cuda", generator=g).to(torch.int32)
    act = torch.randint(0, eng.A, (N,), device="cuda", generator=g).to(torch.int32)
    from hanabizero_amd._lib import poll_giveups
    giveups_before = poll_giveups()
    outs, flags = [], []
    for blockwise, waits in (("1", "1"), ("1", "0"), ("0", "0")):  # counters everywhere | blockwise boundaries only | barriers only
        monkeypatch.setenv("HANABIZERO_MLP_BLOCKWISE", blockwise)
        monkeypatch.setenv("HANABIZERO_MLP_WAITS", waits)
        f = FusedRecurrent(net, eng, 16, 2)
        tab = _job_table(f)
        flags.append([tab[j * 16].flags for j in range(f.n_jobs)])
        h = torch.zeros(N, eng.H, dtype=dtype, device="cuda")
        r, v, p = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, eng.A, device="cuda")
        for _ in range(3):  # (the counters are cleared per launch: a second and third launch must not see stale ones)
            f(pool, ix, act, h, r, v, p, rows_per_wg=rows)
        outs.append((h, r, v, p))
    torch.cuda.synchronize()
    from hanabizero_amd.model import MLP_BARRIER, MLP_WAITS
    assert sum(1 for x in flags[0] if x & MLP_BLOCKWISE) >= 2 and any(x & MLP_SIGNAL for x in flags[0])
    assert any(x & MLP_WAITS for x in flags[0]) and not any(x & MLP_BARRIER for x in flags[0])
    assert not any(x & MLP_WAITS for x in flags[1]) and any(x & MLP_BARRIER for x in flags[1])
    assert not any(x & (MLP_BLOCKWISE | MLP_SIGNAL | MLP_WAITS) for x in flags[2])
    from hanabizero_amd._lib import poll_giveups
    assert poll_giveups() == giveups_before
    bits = lambda t: t.view(torch.int16) if t.dtype != torch.float32 else t
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(bits(a), bits(b))


@pytest.mark.gpu
def test_a_job_table_that_breaks_the_counter_contract_times_out_and_says_so():
    """A consumer that waits for a producer which never signals (the table lies) must not hang the GPU: every wait gives up
    after 2^16 looks, the launch ends, and hz_mlp_poll_giveups counts what happened (include/hz_mlp.h)."""
    import ctypes
    import time
    from hanabizero_amd._lib import MlpJob, poll_giveups
    from hanabizero_amd.model import FusedRecurrent, InferenceEngine, MLP_BLOCKWISE, MLP_SIGNAL
    net, fx, sup = build("Hanabi-Small")
    eng = InferenceEngine(net, sup, dtype=torch.bfloat16, device="cuda")
    f = FusedRecurrent(net, eng, 16, 2)
    tab = _job_table(f)
    first = next(p for p in range(f.n_jobs) if tab[p * 16].flags & MLP_BLOCKWISE)
    for w in range(16):  # the producer of the first blockwise pass forgets to signal
        tab[(first - 1) * 16 + w].flags &= ~MLP_SIGNAL
    f.jobs = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to("cuda")
    N = 16  # one workgroup
    pool = torch.rand(1, N, eng.H, device="cuda").to(torch.bfloat16)
    ix, act = torch.zeros(N, dtype=torch.int32, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda")
    h = torch.zeros(N, eng.H, dtype=torch.bfloat16, device="cuda")
    r, v, p = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, eng.A, device="cuda")
    before = poll_giveups()
    t0 = time.time()
    f(pool, ix, act, h, r, v, p, rows_per_wg=16)
    torch.cuda.synchronize()
    assert time.time() - t0 < 30.0
    assert poll_giveups() > before


@pytest.mark.gpu
@pytest.mark.parametrize("N", [100, 4096])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_initial_tail_matches_gemm_path(N, dtype):
    """FusedInitialTail (the small-GEMM tail of initial_inference in one MFMA launch) against the layer-by-layer bf16
    GEMM path it replaces (same rounding points) and the fp32 engine."""
    from hanabizero_amd.model import InferenceEngine
    net, fx, sup = build("Hanabi-Full")
    eng = InferenceEngine(net, sup, dtype=dtype, device="cuda")
    assert eng.fused_tail is not None
    g = torch.Generator(device="cuda").manual_seed(N)
    D = int(fx["D"]) * int(fx["stack"])
    obs = (torch.rand(N, D, device="cuda", generator=g) < 0.15).to(dtype)  # sparse 0/1 rows like the encoder's
    v, p, h = eng.initial(obs)
    tail, eng.fused_tail = eng.fused_tail, None
    v_ref, p_ref, h_ref = eng.initial(obs)
    eng.fused_tail = tail
    e32 = InferenceEngine(net, sup, dtype=torch.float32, device="cuda")
    v32, p32, h32 = e32.initial(obs.float())

    def mean_err(a, b):
        return float(((a.float() - b.float()).abs() / b.float().abs().clamp(min=1.0)).mean())
    for name, got, ref, truth in [("hidden", h, h_ref, h32), ("policy", p, p_ref, p32), ("value", v, v_ref, v32)]:
        assert got.shape == ref.shape and torch.isfinite(got.float()).all()
        assert mean_err(got, truth) <= 1.3 * mean_err(ref, truth) + 1e-4, (name, mean_err(got, truth), mean_err(ref, truth))
    assert float((h.float() - h32).abs().mean()) < (6e-3 if dtype == torch.bfloat16 else 1e-3)
