"""CPU: the plain-C tree restatement (oracle/tree_oracle.c) against the golden vectors produced by the compiled
reference core/ctree (tools/gen_golden.py), and -- when oracle/_ref is present -- against the reference directly."""
import numpy as np
import pytest

from oracle.cport import OracleTree
from oracle.ref import RefTree, ref_available
from tests.scenarios import bits, load_tree, run_tree_fixture, tree_fixtures


@pytest.mark.parametrize("name", tree_fixtures())
def test_oracle_tree_matches_golden(name):
    fx = load_tree(name)
    run_tree_fixture(lambda N, A, S, seed, delta: OracleTree(N, A, S, seed=seed, value_delta_max=delta), fx)


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("N,A,S,seed", [(48, 20, 50, 11), (24, 11, 50, 12), (8, 48, 50, 13), (16, 20, 20, 14)])
def test_oracle_tree_matches_reference_random(N, A, S, seed):
    rng = np.random.RandomState(seed)
    R, O = RefTree(N, A, S, mode=1, seed=seed), OracleTree(N, A, S, seed=seed)
    noises = rng.dirichlet([0.3] * A, N).astype(np.float32)
    logits = (rng.randn(N, A) * 3).astype(np.float32)
    legal = (rng.rand(N, A) < 0.5).astype(np.int32)
    legal[:, rng.randint(A)] = 1
    for t in (R, O):
        t.prepare(0.25, noises, np.zeros(N), logits, legal)
    for sim in range(S - 1):
        for x, y in zip(R.traverse(sim, 19652, 1.25, 0.999), O.traverse(sim, 19652, 1.25, 0.999)):
            assert (x == y).all()
        r = rng.randint(-2, 3, N).astype(np.float32)
        v = (rng.randn(N) * 10).astype(np.float32)
        l = (rng.randn(N, A) * 2).astype(np.float32)
        R.backprop(sim + 1, 0.999, r, v, l), O.backprop(sim + 1, 0.999, r, v, l)
    assert (R.distributions() == O.distributions()).all()
    assert (bits(R.values()) == bits(O.values())).all()


def test_tree_id_base_shifts_tiebreak_stream():
    """trees [4..8) of one handle == trees [0..4) of a handle created with tree_id_base=4 (multi-GPU sharding)."""
    N, A, S = 8, 11, 20
    rng = np.random.RandomState(0)
    noises = rng.dirichlet([0.3] * A, N).astype(np.float32)
    legal = np.ones((N, A), np.int32)
    z = np.zeros((N, A), np.float32)
    full, shard = OracleTree(N, A, S, seed=9), OracleTree(4, A, S, seed=9, tree_id_base=4)
    full.prepare(0.25, noises, np.zeros(N), z, legal)
    shard.prepare(0.25, noises[4:], np.zeros(4), z[4:], legal[4:])
    for sim in range(S - 1):
        a, b = full.traverse(sim, 19652, 1.25, 0.999), shard.traverse(sim, 19652, 1.25, 0.999)
        assert (a[0][4:] == b[0]).all() and (a[2][4:] == b[2]).all()
        full.backprop(sim + 1, 0.999, np.zeros(N), np.zeros(N), z)
        shard.backprop(sim + 1, 0.999, np.zeros(4), np.zeros(4), z[4:])
    assert (full.distributions()[4:] == shard.distributions()).all()
