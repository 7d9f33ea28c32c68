#!/usr/bin/env python3
"""tools/gen_golden.py -- generate tests/golden/*.npz from the GENUINE reference (oracle/_ref/*.so).

Runs only in the authoring container (needs /root/reference to build oracle/_ref via `make -C oracle ref`).
The fixtures are DATA: inputs and the reference's outputs.  No reference source travels with them.

  tree_<name>.npz   inputs (noises, root logits, legal, fake-net tables) and, per simulation, the reference's
                    (ix, iy, last_action), path length, min/max stats; final visit distributions, root values,
                    greedy trajectories and root priors.       [core/ctree via oracle/ref_tree_harness.cpp]
  env_<game>.npz    per (seed, policy) episode streams: actions and the reference's reward/done/score/state
                    probe/legal mask/share_obs bits after reset and after every step, >=3 episodes per game
                    object so the per-game mt19937 carries across reset.        [envs/hanabi via C API]
  nets_<game>.npz   reference MuZeroNet / MuZeroNetFull (config/hanabi_control/model.py, loaded by file path)
                    state_dict + inputs + initial/recurrent inference outputs, CPU fp32, eval mode.

  search_<game>_autocast.npz   512 roots searched by the reference's nets + tree in fp32 and under fp16 autocast (visit counts,
                    root values): the search-level yardstick of the 16-bit engines.

Usage: python tools/gen_golden.py [--only tree|env|nets|nets_autocast|search_autocast|cpu_rates]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref import RefTree, RefHanabiEnv, ref_available, _hanabi, _Handle  # noqa: E402
import ctypes as C  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
PB_C_BASE, PB_C_INIT, DISCOUNT, DELTA, FRAC = 19652, 1.25, 0.999, 0.006, 0.25  # core/config.py:107-111


# ------------------------------------------------------------------------------------------- tree
def gen_tree(name, N, A, S, family, seed, tie_seed):
    rng = np.random.RandomState(seed)
    noises = rng.dirichlet([0.3] * A, N).astype(np.float32)
    legal = (rng.rand(N, A) < 0.6).astype(np.int32)
    legal[np.arange(N), rng.randint(0, A, N)] = 1
    if family == "all_legal":
        legal[:] = 1
    root_logits = rng.randn(N, A).astype(np.float32)
    root_rewards = np.zeros(N, np.float32)  # core/model.py:71
    rewards = rng.randint(-3, 4, (S - 1, N)).astype(np.float32) * (rng.rand(S - 1, N) < 0.3)
    values = (rng.rand(S - 1, N) * 25).astype(np.float32)
    logits = rng.randn(S - 1, N, A).astype(np.float32)
    with_noise = 1
    if family == "init_zero":  # zero-initialised heads: config/hanabi_control/model.py:151-156
        root_logits[:] = 0
        rewards[:] = 0
        values[:] = 0
        logits[:] = 0
    elif family == "nan":
        root_logits[rng.rand(N, A) < 0.1] = np.nan
        root_logits[0, :] = np.nan
        logits[rng.rand(S - 1, N, A) < 0.02] = np.nan
        values[rng.rand(S - 1, N) < 0.05] = 0
    elif family == "no_noise":  # core/test.py:93 prepare_no_noise
        with_noise = 0
    elif family == "small_delta":  # max-min < value_delta_max branch of cminimax.cpp:35-37
        rewards[:] = 0
        values = (1.0 + rng.rand(S - 1, N) * 0.004).astype(np.float32)
    elif family == "reanalyze":  # reanalyze_worker.py:344-346: noise pre-masked by legal, value prefix as reward
        noises = noises * legal
        root_rewards = rng.randn(N).astype(np.float32)
    rewards = rewards.astype(np.float32)

    t = RefTree(N, A, S, mode=1, seed=tie_seed, value_delta_max=DELTA)
    if with_noise:
        t.prepare(FRAC, noises, root_rewards, root_logits, legal)
    else:
        t.prepare_no_noise(root_rewards, root_logits, legal)
    priors = t.root_priors()
    ixs, iys, las, plens, mins, maxs = [], [], [], [], [], []
    for sim in range(S - 1):  # core/mcts.py:24-26: the last simulation is skipped
        ix, iy, la = t.traverse(sim, PB_C_BASE, PB_C_INIT, DISCOUNT)
        ixs.append(ix), iys.append(iy), las.append(la), plens.append(t.path_len())
        t.backprop(sim + 1, DISCOUNT, rewards[sim], values[sim], logits[sim])
        mn, mx = t.minmax()
        mins.append(mn), maxs.append(mx)
    out = dict(N=N, A=A, S=S, family=family, tie_seed=np.uint64(tie_seed), with_noise=with_noise,
               pb_c_base=PB_C_BASE, pb_c_init=np.float32(PB_C_INIT), discount=np.float32(DISCOUNT),
               value_delta_max=np.float32(DELTA), frac=np.float32(FRAC),
               noises=noises, root_logits=root_logits, root_rewards=root_rewards, legal=legal,
               rewards=rewards, values=values, logits=logits,
               out_root_priors=priors, out_ix=np.array(ixs), out_iy=np.array(iys), out_last_action=np.array(las),
               out_path_len=np.array(plens), out_min=np.array(mins), out_max=np.array(maxs),
               out_distributions=t.distributions(), out_values=t.values(), out_trajectories=t.trajectories(S))
    np.savez_compressed(os.path.join(GOLD, "tree_%s.npz" % name), **out)
    print("tree_%s: N=%d A=%d S=%d family=%s mean path len %.2f" % (name, N, A, S, family, np.mean(plens)))


def check_tree_equivalence():
    """mode 0 (one N-root CRoots, as cytree builds it) == mode 2 (N single-root CRoots), both with rand()==0."""
    N, A, S = 64, 20, 50
    rng = np.random.RandomState(3)
    noises = rng.dirichlet([0.3] * A, N).astype(np.float32)
    legal = (rng.rand(N, A) < 0.6).astype(np.int32)
    legal[:, 0] = 1
    logits0 = rng.randn(N, A).astype(np.float32)
    a, b = RefTree(N, A, S, mode=0), RefTree(N, A, S, mode=2)
    for t in (a, b):
        t.prepare(FRAC, noises, np.zeros(N), logits0, legal)
    for sim in range(S - 1):
        ra, rb = a.traverse(sim, PB_C_BASE, PB_C_INIT, DISCOUNT), b.traverse(sim, PB_C_BASE, PB_C_INIT, DISCOUNT)
        for x, y in zip(ra, rb):
            assert (x == y).all()
        r, v, l = rng.rand(N).astype(np.float32), (rng.rand(N) * 9).astype(np.float32), rng.randn(N, A).astype(np.float32)
        a.backprop(sim + 1, DISCOUNT, r, v, l), b.backprop(sim + 1, DISCOUNT, r, v, l)
    assert (a.distributions() == b.distributions()).all()
    assert (a.values().view(np.uint32) == b.values().view(np.uint32)).all()
    print("reference: one N-root CRoots == N single-root CRoots (rand()==0): OK")


# -------------------------------------------------------------------------------------------- env
class _Card(C.Structure):
    _fields_ = [("color", C.c_int), ("rank", C.c_int)]


def ref_hands(env):
    lib = _hanabi()
    lib.StateGetHandCard.argtypes = [C.POINTER(_Handle), C.c_int, C.c_int, C.POINTER(_Card)]
    hands = []
    for p in range(env.players):
        n = lib.StateGetHandSize(C.byref(env.state), p)
        cards = []
        for i in range(n):
            c = _Card()
            lib.StateGetHandCard(C.byref(env.state), p, i, C.byref(c))
            cards.append((c.color, c.rank))
        hands.append(cards)
    return hands


def policy_action(policy, env, legal, rng, step):
    ids = np.nonzero(legal)[0]
    if policy == "first":
        return int(ids[0])
    if policy == "last":
        return int(ids[-1])
    if policy == "hash":
        return int(ids[(step * 2654435761 + 12345) % len(ids)])
    if policy == "random":
        return int(rng.choice(ids))
    # "smart" / "perfect": look at the true state (the fixture stores the resulting actions, tests only replay them)
    pr = env.probe()
    cur, fw = pr["cur_player"], pr["fireworks"]
    hand = ref_hands(env)[cur]
    hand_size = env.hand_size
    playable = [i for i, (c, r) in enumerate(hand) if fw[c] == r]
    if playable and (policy == "perfect" or rng.rand() < 0.9):
        return hand_size + playable[0]
    dead = [i for i, (c, r) in enumerate(hand) if r < fw[c]]
    hints = [a for a in ids if a >= 2 * hand_size]
    discards = [a for a in ids if a < hand_size]
    if policy == "perfect":
        if discards and dead:
            return dead[0]
        if hints:
            return int(hints[step % len(hints)])
        if discards:
            # discard the card with the highest rank (least likely to be needed soon)
            return int(max(discards, key=lambda a: hand[a][1] if a < len(hand) else -1))
        return int(ids[0])
    if hints and rng.rand() < 0.5:
        return int(rng.choice(hints))
    if discards:
        return int(rng.choice(discards))
    return int(rng.choice(ids))


def gen_env(game, seeds, policies, episodes=3):
    out = {}
    meta = []
    for seed in seeds:
        for policy in policies:
            env = RefHanabiEnv(game, seed)
            rng = np.random.RandomState(seed * 7 + 1)
            acts, rews, dones, scores, probes, legals, obss, boundaries = [], [], [], [], [], [], [], []
            row = 0
            for ep in range(episodes):
                share, _, legal = env.reset()
                p = env.probe()
                boundaries.append(row)
                probes.append([p["cur_player"], p["deck_size"], p["info"], p["life"]] + (p["fireworks"] + [0] * 5)[:5] +
                              (p["hand_sizes"] + [0] * 5)[:5] + [p["status"], p["score"]])
                legals.append(legal.astype(np.uint8)), obss.append(share.astype(np.uint8))
                acts.append(-1), rews.append(0), dones.append(0), scores.append(0)
                row += 1
                step, done = 0, False
                while not done:
                    a = policy_action(policy, env, legal, rng, step)
                    share, _, rew, done, score, legal = env.step(a)
                    p = env.probe()
                    probes.append([p["cur_player"], p["deck_size"], p["info"], p["life"]] + (p["fireworks"] + [0] * 5)[:5] +
                                  (p["hand_sizes"] + [0] * 5)[:5] + [p["status"], p["score"]])
                    legals.append(legal.astype(np.uint8)), obss.append(share.astype(np.uint8))
                    acts.append(a), rews.append(rew), dones.append(int(done)), scores.append(score)
                    row += 1
                    step += 1
            key = "s%d_%s" % (seed, policy)
            out[key + "_action"] = np.array(acts, np.int32)       # -1 marks a reset row
            out[key + "_reward"] = np.array(rews, np.int32)
            out[key + "_done"] = np.array(dones, np.uint8)
            out[key + "_score"] = np.array(scores, np.int32)
            out[key + "_probe"] = np.array(probes, np.int32)
            out[key + "_legal"] = np.packbits(np.array(legals, np.uint8), axis=1)
            out[key + "_obs"] = np.packbits(np.array(obss, np.uint8), axis=1)
            meta.append(key)
            ends = [probes[b - 1][14] for b in boundaries[1:]] + [probes[-1][14]]
            print("env %s %s: %d rows, end statuses %s, final scores %s" % (
                game, key, row, ends, [scores[b - 1] for b in boundaries[1:]] + [scores[-1]]))
    e = RefHanabiEnv(game, 0)
    out["keys"] = np.array(meta)
    out["num_moves"], out["obs_len"], out["own_len"], out["players"] = e.num_moves, e.obs_len, e.own_len, e.players
    np.savez_compressed(os.path.join(GOLD, "env_%s.npz" % game), **out)


# ------------------------------------------------------------------------------------------- nets
def inverse_scalar_transform(logits, support_min, support_max):
    """core/config.py:210-232 restated (delta = 1, epsilon = 0.001); the reference module cannot be imported
    (core/config.py -> core/game.py -> ray)."""
    import torch
    probs = torch.softmax(logits, dim=1)
    support = torch.arange(support_min, support_max + 1, dtype=probs.dtype)
    value = (support * probs).sum(1, keepdim=True)
    eps = 0.001
    sign = torch.ones_like(value)
    sign[value < 0] = -1.0
    out = ((torch.sqrt(1 + 4 * eps * (torch.abs(value) + 1 + eps)) - 1) / (2 * eps)) ** 2 - 1
    out = sign * out
    out[torch.isnan(out)] = 0.0
    return out


def gen_nets():
    import importlib.util
    import torch
    sys.path.insert(0, "/root/reference")
    spec = importlib.util.spec_from_file_location("ref_hanabi_model", "/root/reference/config/hanabi_control/model.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for game, cls, D, A, sup, stack in [("Hanabi-Small", m.MuZeroNet, 193, 11, 25, 1), ("Hanabi-Full", m.MuZeroNetFull, 785, 20, 100, 4)]:
        torch.manual_seed(0)
        inv = lambda x, s=sup: inverse_scalar_transform(x, -s, s)
        net = cls(D * stack, A, 2 * sup + 1, 2 * sup + 1, inv, inv)
        # weights from the shared recipe (tests/netgold.py): non-zero heads, non-trivial BatchNorm statistics
        from tests.netgold import fill_state_dict
        net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
        net.eval()
        B = 32
        obs = (torch.rand(B, D * stack) < 0.3).float()
        with torch.no_grad():
            o0 = net.initial_inference(obs)
            act = torch.randint(0, A, (B, 1))
            o1 = net.recurrent_inference(torch.from_numpy(o0.hidden_state), act)
        out = dict(sd_keys=np.array(list(net.state_dict().keys())))
        out.update(obs=obs.numpy(), action=act.numpy(), init_value=o0.value, init_logits=o0.policy_logits,
                   init_hidden=o0.hidden_state, rec_value=o1.value, rec_reward=o1.reward, rec_logits=o1.policy_logits,
                   rec_hidden=o1.hidden_state, D=D, A=A, support=sup, stack=stack)
        np.savez_compressed(os.path.join(GOLD, "nets_%s.npz" % game), **out)
        print("nets", game, "params", sum(p.numel() for p in net.parameters()))


def gen_nets_autocast():
    """The same reference nets on the same inputs (read back from nets_<game>.npz), run UNMODIFIED under
    torch.autocast(dtype=float16) -- the precision the reference searches with (core/mcts.py:38-40 wraps initial_inference /
    recurrent_inference in autocast(); train.sh:10 --amp_type torch_amp) -- on the CPU, where this torch build supports fp16
    autocast (bf16 autocast computes, but the reference's own `.numpy()` on its outputs refuses bfloat16).  What the fixture pins:
    how far the reference's OWN search-time outputs are from its fp32 outputs, the yardstick for the 16-bit engines."""
    import importlib.util
    import torch
    sys.path.insert(0, "/root/reference")
    spec = importlib.util.spec_from_file_location("ref_hanabi_model", "/root/reference/config/hanabi_control/model.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from tests.netgold import fill_state_dict
    for game, cls in [("Hanabi-Small", m.MuZeroNet), ("Hanabi-Full", m.MuZeroNetFull)]:
        fx = dict(np.load(os.path.join(GOLD, "nets_%s.npz" % game)))
        D, A, sup, stack = int(fx["D"]), int(fx["A"]), int(fx["support"]), int(fx["stack"])
        inv = lambda x, s=sup: inverse_scalar_transform(x.float(), -s, s)
        net = cls(D * stack, A, 2 * sup + 1, 2 * sup + 1, inv, inv)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
        net.eval()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
            o0 = net.initial_inference(torch.from_numpy(fx["obs"]))
            o1 = net.recurrent_inference(torch.from_numpy(fx["init_hidden"]), torch.from_numpy(fx["action"]))
        assert o1.hidden_state.dtype == np.float16, "autocast did not reach the nets"
        out = dict(init_value=o0.value, init_logits=o0.policy_logits, init_hidden=o0.hidden_state, rec_value=o1.value,
                   rec_reward=o1.reward, rec_logits=o1.policy_logits, rec_hidden=o1.hidden_state)
        out = {k: np.asarray(v, np.float32) for k, v in out.items()}
        # a wider sample for statistics that 32 rows cannot carry (the value / reward scalars amplify their logits' error: means
        # over 32 of them are noise): 256 fresh observation windows (bit-packed) and 256 input hidden states that are EXACT in
        # fp16 (the initial inference's fp32 hidden states rounded once: both precisions then see identical inputs, as the search
        # does -- its pool holds 16-bit states), through the fp32 nets and the same nets under fp16 autocast; scalars and policy
        # logits only
        Bw = 256
        g = torch.Generator().manual_seed(1234)
        obs_w = (torch.rand(Bw, D * stack, generator=g) < 0.3)
        act_w = torch.randint(0, A, (Bw, 1), generator=g)
        with torch.no_grad():
            f0 = net.initial_inference(obs_w.float())
            hid_w = torch.from_numpy(f0.hidden_state).half()
            f1 = net.recurrent_inference(hid_w.float(), act_w)
            with torch.autocast("cpu", dtype=torch.float16):
                a0 = net.initial_inference(obs_w.float())
                a1 = net.recurrent_inference(hid_w.float(), act_w)
        out.update(wide_obs_bits=np.packbits(obs_w.numpy(), axis=1), wide_action=act_w.numpy().astype(np.int32),
                   wide_hidden_in=hid_w.numpy())
        for tag, (o0, o1) in (("fp32", (f0, f1)), ("autocast", (a0, a1))):
            out.update({"wide_%s_init_value" % tag: np.asarray(o0.value, np.float32), "wide_%s_init_logits" % tag: np.asarray(o0.policy_logits, np.float32),
                        "wide_%s_rec_value" % tag: np.asarray(o1.value, np.float32), "wide_%s_rec_reward" % tag: np.asarray(o1.reward, np.float32),
                        "wide_%s_rec_logits" % tag: np.asarray(o1.policy_logits, np.float32)})
        np.savez_compressed(os.path.join(GOLD, "nets_%s_autocast.npz" % game), **out)
        out = {k: v for k, v in out.items() if not k.startswith("wide_")}
        worst = max(float(np.max(np.abs(np.asarray(v, np.float64).reshape(-1) - fx[k].astype(np.float64).reshape(-1)) /
                                 np.maximum(1.0, np.abs(fx[k].astype(np.float64).reshape(-1))))) for k, v in out.items())
        print("nets autocast(fp16)", game, "worst |autocast - fp32| / max(1, |fp32|) = %.3g" % worst)


def search_inputs(game, roots, seed):
    """The root set of tests/netgold.py::search_divergence (same generator calls, same order)."""
    D, A, stack = {"Hanabi-Small": (193, 11, 1), "Hanabi-Full": (785, 20, 4)}[game]
    rng = np.random.RandomState(seed)
    obs = (rng.rand(roots, D * stack) < 0.3)
    noise = rng.dirichlet([0.3] * A, roots).astype(np.float32)
    legal = (rng.rand(roots, A) < 0.7).astype(np.uint8)
    legal[:, 0] = 1
    return obs, noise, legal


def gen_search_autocast(roots=512, S=50, seed=0):
    """Search-level yardstick: the same `roots` root positions searched S - 1 simulations by the REFERENCE -- its nets
    (config/hanabi_control/model.py, unmodified) driving its tree (core/ctree through oracle/_ref, tie-breaks bound to
    include/hz_tiebreak.h) with the loop of core/mcts.py:11-57 and the root preparation of core/selfplay_worker.py:268-282 --
    once in fp32 and once under fp16 autocast, the precision the reference searches with (mcts.py:38-40, selfplay_worker.py:
    269-271).  What the fixture pins: how far the reference's OWN fp16 search moves from its fp32 search on these roots (visit
    counts, root values); tests/test_model.py holds the fp16 engine's divergence from the fp32 engine to that."""
    import contextlib
    import importlib.util
    import torch
    sys.path.insert(0, "/root/reference")
    spec = importlib.util.spec_from_file_location("ref_hanabi_model", "/root/reference/config/hanabi_control/model.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    from tests.netgold import fill_state_dict
    for game, cls, sup in [("Hanabi-Small", m.MuZeroNet, 25), ("Hanabi-Full", m.MuZeroNetFull, 100)]:
        obs, noise, legal = search_inputs(game, roots, seed)
        A = noise.shape[1]
        inv = lambda x, s=sup: inverse_scalar_transform(x.float(), -s, s)
        net = cls(obs.shape[1], A, 2 * sup + 1, 2 * sup + 1, inv, inv)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
        net.eval()
        out = dict(obs_bits=np.packbits(obs, axis=1), noise=noise, legal=legal, roots=roots, simulations=S, seed=seed, tie_seed=seed + 1)
        for tag in ("fp32", "autocast"):
            ctx = (lambda: torch.autocast("cpu", dtype=torch.float16)) if tag == "autocast" else contextlib.nullcontext
            with torch.no_grad():
                with ctx():
                    o0 = net.initial_inference(torch.from_numpy(obs).float())             # selfplay_worker.py:269-273
                logits0 = np.asarray(o0.policy_logits, np.float32)
                tree = RefTree(roots, A, S, mode=1, seed=seed + 1, value_delta_max=DELTA)
                tree.prepare(FRAC, noise, np.zeros(roots, np.float32), logits0, legal.astype(np.int32))  # :278-280 (reward_pool = zeros)
                pool = [o0.hidden_state]                                                   # mcts.py:18
                for sim in range(S - 1):                                                   # mcts.py:24-26
                    ix, iy, la = tree.traverse(sim, PB_C_BASE, PB_C_INIT, DISCOUNT)
                    hid = np.asarray([pool[x][y] for x, y in zip(ix, iy)])                 # mcts.py:31-33
                    with ctx():
                        o = net.recurrent_inference(torch.from_numpy(hid), torch.from_numpy(np.asarray(la)).unsqueeze(1).long())
                    lg = np.array(o.policy_logits, np.float32)
                    lg[np.isnan(lg)] = 0.0                                                 # mcts.py:48-49
                    pool.append(o.hidden_state)
                    tree.backprop(sim + 1, DISCOUNT, np.asarray(o.reward, np.float32).reshape(-1), np.asarray(o.value, np.float32).reshape(-1), lg)
                if tag == "autocast":
                    assert pool[-1].dtype == np.float16, "autocast did not reach the nets"
                out["dist_" + tag], out["values_" + tag] = tree.distributions().astype(np.int16), tree.values()
                out["logits0_" + tag] = logits0
        d0, d1 = out["dist_fp32"].astype(np.float64), out["dist_autocast"].astype(np.float64)
        assert (d0.sum(1) == S - 1).all() and (d1.sum(1) == S - 1).all()
        tv = 0.5 * np.abs(d0 - d1).sum(1) / (S - 1)
        print("search autocast(fp16) vs fp32, reference nets + reference tree, %s: same most-visited action %.4f, mean TV %.5f, "
              "identical visit counts %.4f" % (game, (d0.argmax(1) == d1.argmax(1)).mean(), tv.mean(), (tv == 0).mean()))
        np.savez_compressed(os.path.join(GOLD, "search_%s_autocast.npz" % game), **out)


def gen_cpu_rates(N=1024, A=20, S=50, moves=2, env_steps=30):
    """The baseline's baseline: bench.py's `cpu_baseline` times the plain-C PORT
    (oracle/*.c) on the GPU box, where the reference cannot travel.  Here, in the authoring container, the GENUINE reference
    (oracle/_ref: core/ctree/cnode.cpp + cminimax.cpp; envs/hanabi through its C API, driven per env as envs/hanabi/rl_env.py
    drives it) and the port run the same bounded workload on ONE core, so that the port's speed can be read against the thing
    it stands in for.  Written to profiles/r04_cpu_reference_vs_port.json; bench.py attaches it as
    cpu_baseline.reference_in_container with its provenance."""
    import json
    import subprocess
    import time
    from oracle.cport import OracleEnv, OracleTree
    rng = np.random.RandomState(0)
    noises = rng.dirichlet([0.3] * A, N).astype(np.float32)
    logits0 = rng.randn(N, A).astype(np.float32)
    legal = (rng.rand(N, A) < 0.7).astype(np.int32)
    legal[:, 0] = 1
    rew = (rng.randint(-1, 2, (S - 1, N)) * (rng.rand(S - 1, N) < 0.3)).astype(np.float32)
    val = (rng.rand(S - 1, N) * 25).astype(np.float32)
    lg = rng.randn(S - 1, N, A).astype(np.float32)
    zeros = np.zeros(N, np.float32)

    def tree_rate(make):
        best = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            for m in range(moves):
                tree = make()
                tree.prepare(FRAC, noises, zeros, logits0, legal)
                for sim in range(S - 1):
                    tree.traverse(sim, PB_C_BASE, PB_C_INIT, DISCOUNT)
                    tree.backprop(sim + 1, DISCOUNT, rew[sim], val[sim], lg[sim])
                d = tree.distributions()
            best = max(best, N * moves / (time.perf_counter() - t0))
        return best, d
    ref_tree, d_ref = tree_rate(lambda: RefTree(N, A, S, mode=1, seed=0, value_delta_max=DELTA))
    port_tree, d_port = tree_rate(lambda: OracleTree(N, A, S, seed=0, value_delta_max=DELTA))
    assert np.array_equal(d_ref, d_port), "the two searches being timed differ"

    # env: E games, scripted legal actions (the cpu_worker's rule), step + deals + the current player's observation + legal mask
    E = 256
    envs = [RefHanabiEnv("Hanabi-Full", seed=i) for i in range(E)]
    legal_r = np.stack([e.reset()[2] for e in envs])
    t0 = time.perf_counter()
    for m in range(env_steps):
        act = (legal_r * (1 + (np.arange(A) * 7 + m) % A)).argmax(1)
        for i, e in enumerate(envs):
            share, obs, r, done, score, lgl = e.step(int(act[i]))
            if done:
                share, obs, lgl = e.reset()
            legal_r[i] = lgl
    ref_env = E * env_steps / (time.perf_counter() - t0)
    penv = OracleEnv("Hanabi-Full", np.arange(E))
    penv.reset()
    _, legal_p = penv.observe()
    t0 = time.perf_counter()
    for m in range(env_steps * 40):
        act = (legal_p * (1 + (np.arange(A) * 7 + m) % A)).argmax(1).astype(np.int32)
        _, done, _ = penv.step(act)
        if done.any():
            penv.reset(done)
        _, legal_p = penv.observe()
    port_env = E * env_steps * 40 / (time.perf_counter() - t0)
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    out = {"where": "authoring container (no GPU), one core, best of 3 (tree) / one pass (env)", "cpu_model": cpu[0] if cpu else "unknown",
           "commit": head, "workload": "Hanabi-Full 2p: A = %d, %d trees x %d moves x %d simulations; %d envs" % (A, N, moves, S - 1, E),
           "tree_only_root_searches_per_s": {"reference_cnode_cpp": ref_tree, "port_tree_oracle_c": port_tree, "port_over_reference": port_tree / ref_tree,
                                             "note": "both through array-in/array-out C drivers (oracle/ref_tree_harness.cpp, oracle/tree_oracle.c): no Python list conversion, "
                                                     "which is what the reference's Cython binding adds on top (cytree.pyx:87-94)"},
           "env_only_steps_per_s": {"reference_c_api_per_env_via_ctypes": ref_env, "port_env_oracle_c_batched": port_env, "port_over_reference": port_env / ref_env,
                                    "note": "reference: one StateApplyMove + deals + NewObservation + EncodeObservation (ASCII) + legal-move getters per env per step, the call "
                                            "sequence of envs/hanabi/rl_env.py:292-442 for the current player only (rl_env.py encodes ALL players and builds dicts: slower still); "
                                            "port: one C call per batch, bit-packed state"}}
    path = os.path.join(ROOT, "profiles", "r04_cpu_reference_vs_port.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    assert ref_available(), "run `make -C oracle ref` first (needs /root/reference)"
    os.makedirs(GOLD, exist_ok=True)
    if args.only in ("", "tree"):
        check_tree_equivalence()
        gen_tree("small_cfg1", 4, 11, 10, "rand", 0, 0)           # BASELINE config 1 shape
        gen_tree("small_n64", 64, 11, 50, "rand", 1, 0)
        gen_tree("full_n64", 64, 20, 50, "rand", 2, 0)
        gen_tree("full5p_n16", 16, 48, 50, "rand", 3, 0)
        gen_tree("full_init_zero", 64, 20, 50, "init_zero", 4, 0)
        gen_tree("small_init_zero", 32, 11, 50, "init_zero", 5, 7)
        gen_tree("full_nan", 32, 20, 50, "nan", 6, 1)
        gen_tree("full_no_noise", 32, 20, 50, "no_noise", 7, 2)
        gen_tree("full_small_delta", 32, 20, 50, "small_delta", 8, 3)
        gen_tree("full_reanalyze", 32, 20, 50, "reanalyze", 9, 4)
        gen_tree("full_all_legal_s100", 8, 20, 100, "all_legal", 10, 5)
    if args.only in ("", "env"):
        gen_env("Hanabi-Small", [0, 1, 7, 123], ["first", "last", "hash", "smart", "perfect"], episodes=4)
        gen_env("Hanabi-Full", [0, 1, 7, 123], ["first", "last", "hash", "smart", "perfect"], episodes=3)
        gen_env("Hanabi-Full-5p", [0, 1, 7, 123], ["first", "hash", "smart", "perfect"], episodes=3)
    if args.only in ("", "nets"):
        gen_nets()
    if args.only in ("", "nets", "nets_autocast"):
        gen_nets_autocast()
    if args.only in ("", "search_autocast"):
        gen_search_autocast()
    if args.only == "cpu_rates":  # (a timing, not a fixture: only on request)
        gen_cpu_rates()


if __name__ == "__main__":
    main()
