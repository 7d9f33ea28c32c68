"""GPU: the HIP search tree, called through the C ABI, against (i) the golden vectors of the compiled reference,
(ii) the plain-C oracle on seeded random inputs, (iii) size-independent properties at BASELINE.json's full sizes."""
import numpy as np
import pytest
import torch

from tests.scenarios import bits, load_tree, run_tree_fixture, tree_fixtures

pytestmark = pytest.mark.gpu


def make_hip(N, A, S, seed, delta):
    from tests.hip_adapters import HipTree
    return HipTree(N, A, S, seed=seed, value_delta_max=delta)


@pytest.mark.parametrize("name", tree_fixtures())
def test_hip_tree_matches_reference_golden(name):
    run_tree_fixture(make_hip, load_tree(name))


def _random_case(N, A, S, seed, zero=False, scale=1.0):
    rng = np.random.RandomState(seed)
    c = dict(noises=rng.dirichlet([0.3] * A, N).astype(np.float32),
             logits0=(rng.randn(N, A) * 3).astype(np.float32),
             legal=(rng.rand(N, A) < 0.5).astype(np.uint8),
             rewards=(rng.randint(-2, 3, (S - 1, N)) * (rng.rand(S - 1, N) < 0.4)).astype(np.float32),
             values=(rng.rand(S - 1, N) * 25 * scale).astype(np.float32),
             logits=(rng.randn(S - 1, N, A) * 2).astype(np.float32))
    c["legal"][np.arange(N), rng.randint(0, A, N)] = 1
    if zero:
        for k in ("logits0", "rewards", "values", "logits"):
            c[k][:] = 0
    return c


@pytest.mark.parametrize("N,A,S,seed,zero", [(1024, 20, 50, 21, False), (512, 11, 50, 22, False), (256, 48, 50, 23, False),
                                            (1024, 20, 50, 24, True), (3, 20, 50, 25, False), (130, 11, 10, 26, True),
                                            (64, 20, 200, 27, False)])
def test_hip_tree_matches_oracle_random(N, A, S, seed, zero):
    from oracle.cport import OracleTree
    c = _random_case(N, A, S, seed, zero)
    O, H = OracleTree(N, A, S, seed=seed), make_hip(N, A, S, seed, 0.006)
    for t in (O, H):
        t.prepare(0.25, c["noises"], np.zeros(N, np.float32), c["logits0"], c["legal"])
    assert (bits(O.root_priors()) == bits(H.root_priors())).all()
    for sim in range(S - 1):
        a, b = O.traverse(sim, 19652, 1.25, 0.999), H.traverse(sim, 19652, 1.25, 0.999)
        for x, y, nm in zip(a, b, ("ix", "iy", "la")):
            assert (x == y).all(), (nm, sim, np.nonzero(x != y)[0][:5])
        for t in (O, H):
            t.backprop(sim + 1, 0.999, c["rewards"][sim], c["values"][sim], c["logits"][sim])
        (omn, omx), (hmn, hmx) = O.minmax(), H.minmax()
        assert (omn == hmn).all() and (omx == hmx).all(), sim
    assert (O.distributions() == H.distributions()).all()
    assert (bits(O.values()) == bits(H.values())).all()
    assert (O.trajectories() == H.trajectories()).all()


def test_full_size_properties():
    """BASELINE.json full size (4096 trees, A=20, S=50): visit counts sum to S-1, root value = mean of backed-up
    returns, sharding invariance (trees [2048..4096) == a second handle with tree_id_base=2048)."""
    N, A, S = 4096, 20, 50
    c = _random_case(N, A, S, 31)
    full = make_hip(N, A, S, 5, 0.006)
    from tests.hip_adapters import HipTree
    shard = HipTree(N // 2, A, S, seed=5, value_delta_max=0.006, tree_id_base=N // 2)
    full.prepare(0.25, c["noises"], np.zeros(N, np.float32), c["logits0"], c["legal"])
    h = N // 2
    shard.prepare(0.25, c["noises"][h:], np.zeros(h, np.float32), c["logits0"][h:], c["legal"][h:])
    for sim in range(S - 1):
        a, b = full.traverse(sim, 19652, 1.25, 0.999), shard.traverse(sim, 19652, 1.25, 0.999)
        assert (a[0][h:] == b[0]).all() and (a[2][h:] == b[2]).all()
        assert (a[1] == np.arange(N)).all() and (a[0] <= sim).all() and (a[0] >= 0).all()
        full.backprop(sim + 1, 0.999, c["rewards"][sim], c["values"][sim], c["logits"][sim])
        shard.backprop(sim + 1, 0.999, c["rewards"][sim][h:], c["values"][sim][h:], c["logits"][sim][h:])
    d = full.distributions()
    assert (d.sum(1) == S - 1).all() and (d >= 0).all()
    assert (d[h:] == shard.distributions()).all()
    assert np.isfinite(full.values()).all()
    mn, mx = full.minmax()
    assert (mn <= mx).all()


def test_cytree_list_api_matches_reference_signatures():
    """The compatibility path: python lists in, python lists out, exactly the calls core/mcts.py:20-57 makes."""
    from hanabizero_amd import cytree as tree
    fx = load_tree("small_cfg1")
    N, A, S = int(fx["N"]), int(fx["A"]), int(fx["S"])
    roots = tree.Roots(N, A, S, tie_seed=int(fx["tie_seed"]))
    roots.prepare(float(fx["frac"]), fx["noises"].tolist(), fx["root_rewards"].tolist(), fx["root_logits"].tolist(),
                  fx["legal"].tolist())
    mm = tree.MinMaxStatsList(N)
    mm.set_delta(float(fx["value_delta_max"]))
    for sim in range(S - 1):
        results = tree.ResultsWrapper(N)
        ix, iy, la = tree.multi_traverse(roots, int(fx["pb_c_base"]), float(fx["pb_c_init"]), float(fx["discount"]), mm, results)
        assert isinstance(ix, list) and ix == fx["out_ix"][sim].tolist() and la == fx["out_last_action"][sim].tolist()
        tree.multi_back_propagate(sim + 1, float(fx["discount"]), fx["rewards"][sim].tolist(), fx["values"][sim].tolist(),
                                  fx["logits"][sim].tolist(), mm, results)
    assert roots.get_distributions() == fx["out_distributions"].tolist()
    assert np.array_equal(np.float32(roots.get_values()), fx["out_values"])
    assert roots.get_trajectories() == [[a for a in row if a >= 0] for row in fx["out_trajectories"].tolist()]
    assert roots.num == N


def test_errors_are_reported_not_fatal():
    from hanabizero_amd import cytree as tree
    from hanabizero_amd._lib import HzError
    roots = tree.Roots(4, 11, 10)
    z = np.zeros((4, 11), np.float32)
    roots.prepare(0.25, z + 1.0 / 11, np.zeros(4), z, np.ones((4, 11), np.uint8))
    roots.set_params(19652, 1.25, 0.999, 0.006)
    roots.traverse_tensors()
    with pytest.raises(HzError):
        roots.backprop_tensors(3, np.zeros(4), np.zeros(4), z)  # must be 1 after prepare
    with pytest.raises(HzError):
        tree.Roots(4, 65, 10)


def test_traverse_gather_moves_the_right_rows():
    from hanabizero_amd import cytree as tree
    N, A, S, H = 256, 20, 12, 512
    c = _random_case(N, A, S, 41)
    for dtype in (torch.bfloat16, torch.float32):
        roots = tree.Roots(N, A, S, tie_seed=3)
        roots.prepare(0.25, c["noises"], np.zeros(N), c["logits0"], c["legal"])
        roots.set_params(19652, 1.25, 0.999, 0.006)
        pool = torch.randn(S, N, H, device="cuda").to(dtype)
        net_in = torch.zeros(N, H + 32, device="cuda", dtype=dtype)
        for sim in range(S - 1):
            ix, iy, la = roots.traverse_tensors(pool, net_in)
            want = pool[ix.long(), torch.arange(N, device="cuda")]
            assert torch.equal(net_in[:, :H], want)
            assert (net_in[:, H:] == 0).all()
            roots.backprop_tensors(sim + 1, c["rewards"][sim], c["values"][sim], c["logits"][sim])


def test_traverse_gather_writes_one_hot_action():
    from hanabizero_amd import cytree as tree
    N, A, S, H, OH = 128, 20, 8, 512, 32
    c = _random_case(N, A, S, 43)
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        roots = tree.Roots(N, A, S, tie_seed=4)
        roots.prepare(0.25, c["noises"], np.zeros(N), c["logits0"], c["legal"])
        roots.set_params(19652, 1.25, 0.999, 0.006)
        pool = torch.randn(S, N, H, device="cuda").to(dtype)
        net_in = torch.full((N, H + OH), 5.0, device="cuda", dtype=dtype)
        for sim in range(S - 1):
            ix, iy, la = roots.traverse_tensors(pool, net_in, onehot_cols=OH)
            assert torch.equal(net_in[:, :H], pool[ix.long(), torch.arange(N, device="cuda")])
            want = torch.zeros(N, OH, device="cuda", dtype=dtype)
            want[torch.arange(N, device="cuda"), la.long()] = 1
            assert torch.equal(net_in[:, H:], want)
            roots.backprop_tensors(sim + 1, c["rewards"][sim], c["values"][sim], c["logits"][sim])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_backprop_nets_scalar_transform_and_exact_tree(dtype):
    """hz_tree_backprop_nets: (i) the in-kernel scalar transform agrees with core/config.py:210-232 evaluated by
    torch in fp32 to 1e-4 (a net output: north-star tolerance 1e-3); (ii) fed those scalars and the NaN-cleared
    logits, the plain-C oracle ends in the identical tree."""
    from hanabizero_amd import cytree as tree
    from hanabizero_amd.model import inverse_scalar_transform
    from oracle.cport import OracleTree
    N, A, S, V, PAD = 192, 20, 30, 201, 256
    c = _random_case(N, A, S, 44)
    g = torch.Generator(device="cuda").manual_seed(1)
    roots, orc = tree.Roots(N, A, S, tie_seed=6), OracleTree(N, A, S, seed=6)
    roots.prepare(0.25, c["noises"], np.zeros(N), c["logits0"], c["legal"])
    orc.prepare(0.25, c["noises"], np.zeros(N, np.float32), c["logits0"], c["legal"])
    roots.set_params(19652, 1.25, 0.999, 0.006)
    for sim in range(S - 1):
        ix, iy, la = roots.traverse_tensors()
        oix, _, ola = orc.traverse(sim, 19652, 1.25, 0.999)
        assert (ix.cpu().numpy() == oix).all() and (la.cpu().numpy() == ola).all(), sim
        heads = (torch.randn(3, N, PAD, device="cuda", generator=g) * 3).to(dtype)  # rows wider than V / A: strided views
        if sim == 3:
            heads[1, :7, 2] = float("nan")      # NaN policy logits are cleared (core/mcts.py:48-49)
            heads[0, 5, :] = float("nan")       # a NaN reward row becomes 0 (config.py:229-232)
        out_r = torch.empty(N, device="cuda")
        out_v = torch.empty(N, device="cuda")
        roots.backprop_nets_tensors(sim + 1, heads[0], heads[2], V, -100, heads[1], out_r, out_v)
        for got, src in ((out_r, heads[0]), (out_v, heads[2])):
            want = inverse_scalar_transform(src[:, :V].float(), -100, 100).reshape(-1)
            assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
        lg = torch.nan_to_num(heads[1][:, :A].float(), nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        orc.backprop(sim + 1, 0.999, out_r.cpu().numpy(), out_v.cpu().numpy(), lg.cpu().numpy())
        (omn, omx), (hmn, hmx) = orc.minmax(), roots.minmax_tensors()
        assert (omn == hmn.cpu().numpy()).all() and (omx == hmx.cpu().numpy()).all()
    assert (roots.distributions_tensor().cpu().numpy() == orc.distributions()).all()
    assert (bits(roots.values_tensor().cpu().numpy()) == bits(orc.values())).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_add_relu_glue_kernel(dtype):
    from hanabizero_amd._lib import check, lib
    torch.manual_seed(0)
    y = torch.randn(300, 512, device="cuda").to(dtype)
    big = torch.randn(300, 544, device="cuda").to(dtype)
    res = big[:, :512]  # strided residual, like net_in[:, :H]
    want = torch.relu(y + res)
    dt = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dtype]
    check(lib.hz_add_relu(y.data_ptr(), y.stride(0), res.data_ptr(), res.stride(0), 300, 512, dt, torch.cuda.current_stream().cuda_stream), "x")
    assert torch.equal(y, want)


@pytest.mark.parametrize("N,A,S,zero", [(700, 20, 50, False), (300, 11, 50, True), (64, 48, 30, False), (40, 20, 90, False)])
def test_fused_backprop_traverse_equals_separate_calls(N, A, S, zero):
    """hz_tree_backprop_traverse (one launch per simulation) against hz_tree_backprop + hz_tree_traverse and the oracle."""
    from hanabizero_amd import cytree as tree
    from oracle.cport import OracleTree
    c = _random_case(N, A, S, 51, zero)
    a, b = tree.Roots(N, A, S, tie_seed=8), tree.Roots(N, A, S, tie_seed=8)
    orc = OracleTree(N, A, S, seed=8)
    for r in (a, b):
        r.prepare(0.25, c["noises"], np.zeros(N), c["logits0"], c["legal"])
        r.set_params(19652, 1.25, 0.999, 0.006)
    orc.prepare(0.25, c["noises"], np.zeros(N, np.float32), c["logits0"], c["legal"])
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    ia = [t.clone() for t in a.traverse_tensors()]
    ib = [t.clone() for t in b.traverse_tensors()]
    for sim in range(S - 1):
        o = orc.traverse(sim, 19652, 1.25, 0.999)
        for x, y, z in zip(ia, ib, o):
            assert torch.equal(x, y) and (x.cpu().numpy() == z).all(), sim
        assert torch.equal(a.path_len_tensor(), b.path_len_tensor())
        rw, vl, lg = dev(c["rewards"][sim]), dev(c["values"][sim]), dev(c["logits"][sim])
        orc.backprop(sim + 1, 0.999, c["rewards"][sim], c["values"][sim], c["logits"][sim])
        if sim < S - 2:
            a.backprop_tensors(sim + 1, rw, vl, lg)
            ia = [t.clone() for t in a.traverse_tensors()]
            ib = [t.clone() for t in b.backprop_traverse_tensors(sim + 1, rw, vl, lg)]
        else:
            a.backprop_tensors(sim + 1, rw, vl, lg)
            b.backprop_tensors(sim + 1, rw, vl, lg)
        for x, y in zip(a.minmax_tensors(), b.minmax_tensors()):
            assert torch.equal(x, y)
    assert torch.equal(a.distributions_tensor(), b.distributions_tensor())
    assert (b.distributions_tensor().cpu().numpy() == orc.distributions()).all()
    assert torch.equal(a.values_tensor(), b.values_tensor())
    assert torch.equal(a.trajectories_tensor(), b.trajectories_tensor())
