#!/usr/bin/env python3
"""bench.py -- self-play moves/sec & MCTS sims/sec of the MI355X engine on BASELINE.json's headline workload.

One "step" = one lock-step of the self-play hot path over every env of this GPU: root inference -> root prepare ->
49 x (HIP select + gather -> dynamics/prediction nets -> HIP expand/backup) [one persistent kernel] -> read-out -> action
sampling -> HIP env step + encode -> finished-game flush -> reset.  Inputs are resident in HBM; nothing is skipped.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Besides the driver's contract fields it carries
  net_error     measured error of the nets in the dtype of this run against the reference nets' fp32 outputs (tests/golden), against
                the reference run under fp16 autocast (its own search precision), and what it does to the search's visit counts
  roofline      the dominant hand-written kernel (k_search): arithmetic per launch / MEAN launch duration measured here with
                HIP events on the launch stream, against the dense MFMA peak; beside it the L1<-L2 weight-stream rate
                against the L2 peak (the binding resource), the min launch duration, and `traffic` = HBM bytes per launch
                from the committed PMC passes (profiles/pmc_traffic.json, with the commit they were taken at)
  cpu_baseline  the plain-C oracle (tree + env, no nets: the part the reference runs on CPU) on the GPU box's host cores:
                one core on a bounded sample, tree-only / env-only splits, and one process per core.  A baseline, not the target.
  also          (N = 1) other configurations measured in the same invocation: BASELINE configs[2] (8192 envs; in the run's format and
                in the bf16 that config names), the same workload with bf16 nets, and a sharp-policy net (deep search paths)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (game, envs per GPU, simulations, stack)
    "full4096": ("Hanabi-Full", 4096, 50, 4),     # BASELINE.json metric: Hanabi-Full 2p, 50 sims, 4096 envs
    "small4096": ("Hanabi-Small", 4096, 50, 4),   # BASELINE.json configs[1]
    "full8192": ("Hanabi-Full", 8192, 50, 4),     # BASELINE.json configs[2] (and [3] at --gpus 8)
    "full16384": ("Hanabi-Full", 16384, 50, 4),   # scaling probes beyond the named configs (288 GB HBM has room)
    "full32768": ("Hanabi-Full", 32768, 50, 4),
    "full5p2048": ("Hanabi-Full-5p", 2048, 50, 4),  # BASELINE.json configs[4]'s game (A = 48, D = 1385), self-play part
    "full5p4096": ("Hanabi-Full-5p", 4096, 50, 4),  # ... with a workgroup of 16 trees on every compute unit
    "full5p8192": ("Hanabi-Full-5p", 8192, 50, 4),  # ... and with 32 trees per workgroup, one after the other (A > 32: k_search_turn)
}
HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0       # same guide: ~2.5 PFLOP/s dense bf16 (fp16 runs at the same rate)
L2_PEAK_TBS = 34.5              # same guide, section L2: ~34.5 TB/s aggregate over the 8 XCDs


# ---------------------------------------------------------------------------------------------------- CPU baseline
def cpu_worker(kind, game, A, S, trees, moves):
    """Oracle tree and / or oracle env on ONE host core (this process): per move prepare + (S-1) x (traverse, backprop with
    recorded fake-net outputs) [tree] and env step + encode [env].  No torch, no GPU.  Returns (moves/s, seconds)."""
    import numpy as np
    from oracle.cport import OracleEnv, OracleTree
    rng = np.random.RandomState(0)
    N = trees
    if kind == "ref_env":   # the GENUINE rules engine + encoder (oracle/_ref/libpyhanabi.so), one object per game as envs/hanabi/rl_env.py drives it
        from oracle.ref import RefHanabiEnv
        envs = [RefHanabiEnv(game, seed=i) for i in range(N)]
        legal = np.stack([e.reset()[2] for e in envs])
        t0 = time.perf_counter()
        for m in range(moves):
            act = (legal * (1 + (np.arange(A) * 7 + m) % A)).argmax(1)
            for i, e in enumerate(envs):
                out = e.step(int(act[i]))
                legal[i] = e.reset()[2] if out[3] else out[5]
        dt = time.perf_counter() - t0
        return N * moves / dt, dt
    env = OracleEnv(game, np.arange(N))
    env.reset()
    obs, legal = env.observe()
    noises = rng.dirichlet([0.3] * A, N).astype(np.float32)
    logits0 = rng.randn(N, A).astype(np.float32)
    rew = (rng.randint(-1, 2, (S - 1, N)) * (rng.rand(S - 1, N) < 0.3)).astype(np.float32)
    val = (rng.rand(S - 1, N) * 25).astype(np.float32)
    lg = rng.randn(S - 1, N, A).astype(np.float32)
    zeros = np.zeros(N, np.float32)
    t0 = time.perf_counter()
    for m in range(moves):
        if kind in ("both", "tree", "ref_tree"):
            if kind == "ref_tree":   # the GENUINE tree (oracle/_ref/libref_tree.so: core/ctree/cnode.cpp + cminimax.cpp)
                from oracle.ref import RefTree
                tree = RefTree(N, A, S, mode=1, seed=0, value_delta_max=0.006)
            else:
                tree = OracleTree(N, A, S, seed=0, value_delta_max=0.006)  # the reference builds a new Roots per move
            tree.prepare(0.25, noises, zeros, logits0, legal)
            for sim in range(S - 1):
                tree.traverse(sim, 19652, 1.25, 0.999)
                tree.backprop(sim + 1, 0.999, rew[sim], val[sim], lg[sim])
            dist = tree.distributions().astype(np.float64) * legal
            act = (dist + 1e-3 * legal).argmax(1).astype(np.int32)
        else:  # scripted legal actions
            act = (legal * (1 + (np.arange(A) * 7 + m) % A)).argmax(1).astype(np.int32)
        if kind in ("both", "env"):
            _, done, _ = env.step(act)
            if done.any():
                env.reset(done)
            obs, legal = env.observe()
    dt = time.perf_counter() - t0
    return N * moves / dt, dt


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(game, A, S, trees, moves, workload, max_procs):
    """All legs as child processes that never import torch or touch the GPU (python bench.py --cpu-worker ...)."""
    def spawn(kind, t, m):
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", kind, "--workload", workload,
                                 "--cpu-sample-trees", str(t), "--cpu-sample-moves", str(m)], stdout=subprocess.PIPE, text=True)

    def result(p):
        out, _ = p.communicate()
        return json.loads(out.strip().splitlines()[-1])
    one = result(spawn("both", trees, moves))
    tree = result(spawn("tree", trees, max(1, moves // 2)))
    env = result(spawn("env", trees, moves * 12))
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(avail, max_procs))
    per = max(256, trees // 2)
    t0 = time.perf_counter()
    rs = [result(p) for p in [spawn("both", per, max(2, moves // 2)) for _ in range(procs)]]
    wall = time.perf_counter() - t0
    here = None
    try:  # ... and the genuine reference's own tree and env on THIS box's cores where its build travelled with the snapshot (oracle/_ref)
        ref_dir = os.path.join(ROOT, "oracle", "_ref")   # (the children load it: this process never imports anything under oracle/)
        if os.path.exists(os.path.join(ref_dir, "libref_tree.so")) and os.path.exists(os.path.join(ref_dir, "libpyhanabi.so")):
            rt, re_ = result(spawn("ref_tree", min(trees, 1024), 2)), result(spawn("ref_env", 256, 30))
            here = {"kind": "reference", "what": "oracle/_ref: the reference's core/ctree/cnode.cpp + cminimax.cpp and envs/hanabi (hanabi_lib + pyhanabi.cc) compiled as they lie, driven through ctypes",
                    "tree_only": {"value": rt["moves_per_s"], "unit": "root-searches/s", "cores": 1, "sample": "%d trees x 2 moves x %d sims, %.1f s" % (min(trees, 1024), S - 1, rt["seconds"])},
                    "env_only": {"value": re_["moves_per_s"], "unit": "env-steps/s (step + observation + legal moves, one C-API object per game)", "cores": 1,
                                 "sample": "256 games x 30 moves, %.1f s" % re_["seconds"]}}
    except Exception as e:  # noqa: BLE001
        here = {"error": repr(e)[:300]}
    ref = None
    try:  # the genuine reference beside the port, timed where the reference exists (tools/gen_golden.py --only cpu_rates); a pointer
        # with its provenance, like roofline.traffic_source -- nothing of it is measured in this run
        ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_cpu_reference_vs_port.json")))
        ref["source"] = "profiles/r04_cpu_reference_vs_port.json"
    except (OSError, ValueError):
        pass
    return {"value": one["moves_per_s"], "unit": "moves/s", "cores": 1, "kind": "port", "reference_on_this_box": here, "reference_in_container": ref,
            "sample": "oracle tree+env (no nets), %d envs x %d moves x %d sims, %.1f s on 1 core" % (trees, moves, S - 1, one["seconds"]),
            "cpu_model": cpu_model(), "machine_cores": os.cpu_count(), "cores_available": avail,
            "tree_only": {"value": tree["moves_per_s"], "unit": "root-searches/s", "cores": 1,
                          "sample": "%d trees x %d moves x %d sims, %.1f s" % (trees, max(1, moves // 2), S - 1, tree["seconds"])},
            "env_only": {"value": env["moves_per_s"], "unit": "env-steps/s (step + deal + encode)", "cores": 1,
                         "sample": "%d envs x %d moves, %.1f s" % (trees, moves * 12, env["seconds"])},
            "all_cores": {"value": sum(r["moves_per_s"] for r in rs), "unit": "moves/s", "cores": procs,
                          "sample": "%d processes x (%d envs x %d moves), one per core, %.1f s wall" % (procs, per, max(2, moves // 2), wall)}}


# ---------------------------------------------------------------------------------------------------- engine
def build_engine(cfg, dtype, device, fused=None, net="random"):
    """net: "random" = SURVEY 8d's fixed random init with the zero-initialised heads perturbed by N(0, 0.1);
    "sharp" = the same with the policy head's output layer scaled up, so that priors concentrate on few actions per node the way
    a trained policy's do and the search goes deep (the reference's tree walk: core/ctree/cnode.cpp:407-441)."""
    import torch
    from hanabizero_amd.model import InferenceEngine
    torch.manual_seed(0)
    network = cfg.get_uniform_network()
    with torch.no_grad():
        for head in (network._prediction_value, network._dynamics_reward, network._prediction_actor):
            head[-1].weight.normal_(0, 0.1)
            head[-1].bias.normal_(0, 0.1)
        if net.startswith("sharp"):
            scale = float(net.split(":")[1]) if ":" in net else 40.0
            network._prediction_actor[-1].weight.mul_(scale)
            network._prediction_actor[-1].bias.mul_(scale)
    network.eval()
    return InferenceEngine(network, cfg.value_support.max, dtype=dtype, device=device, fused=fused)


def mean_path_edges(actor):
    """Mean root -> leaf path length (edges) over ALL simulations of one search of the actor's current position
    (launch-per-phase search; leaves the actor's own state alone except its scratch trees)."""
    import torch
    cfg, roots, eng = actor.cfg, actor.roots, actor.engine
    N, S = actor.N, actor.S
    actor._draw()
    actor._drawn = True
    value0, logits0, hidden0 = actor.root_inference()
    roots.prepare(cfg.root_exploration_fraction, actor.noise, actor.zeros_n, logits0, actor.legal)
    roots.set_params(cfg.pb_c_base, cfg.pb_c_init, cfg.discount, cfg.value_delta_max)
    pool = torch.empty_like(actor.pool)
    pool[0].copy_(hidden0)
    rew = torch.empty(N, dtype=torch.float32, device=actor.device)
    val = torch.empty(N, dtype=torch.float32, device=actor.device)
    pol = torch.empty((N, actor.A), dtype=torch.float32, device=actor.device)
    total = torch.zeros((), dtype=torch.float64, device=actor.device)
    deepest = 0
    ix, _, la = roots.traverse_tensors()
    for sim in range(S - 1):
        pl = roots.path_len_tensor()
        total += pl.double().mean() - 1.0
        deepest = max(deepest, int(pl.max()) - 1)
        eng.fused(pool, ix, la, pool[sim + 1], rew, val, pol)
        if sim < S - 2:
            ix, _, la = roots.backprop_traverse_tensors(sim + 1, rew, val, pol)
        else:
            roots.backprop_tensors(sim + 1, rew, val, pol)
    return float(total) / (S - 1), deepest


def search_in_step(actor, steps=32):
    """Launch duration of the persistent search kernel INSIDE lock-steps: `steps` more moves of the live actor, enqueued
    kernel by kernel (the captured graph's own sequence: root inference, prepare, search, move tail) with a HIP event on
    either side of the search launch, on the launch stream.  The stream stays full (the host runs ahead of a 1.9 ms step), so a
    pair brackets the kernel and nothing else; an empty pair recorded right behind it gives the events' own share (~5 us, reported,
    not subtracted).  Until r03 this figure came from graphs of four back-to-back launches on snapshots: 8-11 % longer than the
    same kernel takes between the GEMMs and the move tail of a real step (no kernel before it to rest the clocks, cold trees)."""
    import torch
    cfg = actor.cfg
    ev = lambda: torch.cuda.Event(enable_timing=True)
    pairs, empties = [], []
    first, last = ev(), ev()
    warm = 8  # (the first kernel-by-kernel steps of a process that has replayed graphs so far are slow on the HOST: not timed)
    for k in range(warm + steps):
        if k == warm:
            pairs, empties = [], []
            first.record()
        if not actor._drawn:
            actor._draw()
        value0, logits0, hidden0 = actor.root_inference(state_out=actor.pool[0])
        actor.roots.prepare(cfg.root_exploration_fraction, actor.noise, actor.zeros_n, logits0, actor.legal)
        a, b, c, d = ev(), ev(), ev(), ev()
        a.record()
        actor.mcts.run_multi(actor.roots, actor.engine, hidden0, pool=actor.pool)
        b.record()
        c.record()
        d.record()
        actor._tail_part(True)
        actor.total_moves += actor.N
        pairs.append((a, b))
        empties.append((c, d))
    last.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(b) * 1e-3 for a, b in pairs]
    e = [c.elapsed_time(d) * 1e-3 for c, d in empties]
    return {"mean_s": sum(t) / len(t), "min_s": min(t), "launches": len(t), "empty_event_pair_us": 1e6 * sum(e) / len(e),
            "eager_step_us": 1e3 * first.elapsed_time(last) / steps}


def kernel_timing(actor, sample_sims=(4, 16, 28, 40), clones=8, replays=5):
    """Launch durations of the hand-written kernels on LIVE search states, with HIP events on the launch stream.

    k_search (the kernel the product launches once per move): inside further lock-steps of the live actor (search_in_step).
    The launch-per-phase kernels: hipGraphs of `clones` back-to-back launches on snapshots (hz_tree_copy) at each sampled
    simulation, between two events, best of `replays` (an eager launch of a 10 us kernel cannot be timed with an event pair).
    Returns ({kernel: seconds per launch}, {k_search statistics}, launches, mean path edges, mean expanded entries)."""
    import torch
    cfg, roots, eng = actor.cfg, actor.roots, actor.engine
    N, S, oh = actor.N, actor.S, eng.onehot_cols
    actor._draw()
    actor._drawn = True
    value0, logits0, hidden0 = actor.root_inference()
    roots.prepare(cfg.root_exploration_fraction, actor.noise, actor.zeros_n, logits0, actor.legal)
    roots.set_params(cfg.pb_c_base, cfg.pb_c_init, cfg.discount, cfg.value_delta_max)
    actor.pool[0].copy_(hidden0)
    net_in = torch.empty((N, eng.H + oh), dtype=eng.dtype, device=actor.device)
    fused = getattr(eng, "fused", None)
    tot = {"k_traverse": 0.0, "k_backprop": 0.0, "k_mlp_recurrent": 0.0, "k_backprop_traverse": 0.0}
    rew = torch.empty(N, dtype=torch.float32, device=actor.device)
    val = torch.empty(N, dtype=torch.float32, device=actor.device)
    pol = torch.empty((N, actor.A), dtype=torch.float32, device=actor.device)
    launches, depth, entries = 0, 0.0, 0.0
    ev = lambda: torch.cuda.Event(enable_timing=True)
    search = None

    def timed_graph(body):
        side = torch.cuda.Stream(device=actor.device)
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            body()
        return g

    fused16 = eng.fused_shape(16, 2) if (fused is not None and getattr(actor.mcts, "persistent", False)) else None
    if fused16 is not None:
        search = search_in_step(actor)
        # (the in-step loop has moved the games on: prepare this function's own roots again for what follows)
        actor._draw()
        actor._drawn = True
        value0, logits0, hidden0 = actor.root_inference()
        roots.prepare(cfg.root_exploration_fraction, actor.noise, actor.zeros_n, logits0, actor.legal)
        roots.set_params(cfg.pb_c_base, cfg.pb_c_init, cfg.discount, cfg.value_delta_max)
        actor.pool[0].copy_(hidden0)
    if sample_sims:
        for sim in range(S - 1):
            sampled = sim in sample_sims
            if sampled:
                snaps = [roots.clone() for _ in range(clones)]
                ins = [torch.empty_like(net_in) for _ in range(clones)]
                torch.cuda.synchronize()
                if fused is not None:
                    g = timed_graph(lambda: [c.traverse_tensors() for c in snaps])
                else:
                    g = timed_graph(lambda: [c.traverse_tensors(actor.pool, b, onehot_cols=oh) for c, b in zip(snaps, ins)])
            if fused is not None:
                roots.traverse_tensors()
            else:
                roots.traverse_tensors(actor.pool, net_in, onehot_cols=oh)
            if sampled:
                best = 1e9
                for _ in range(replays):
                    a, b = ev(), ev()
                    a.record(); g.replay(); b.record()
                    torch.cuda.synchronize()
                    best = min(best, a.elapsed_time(b) * 1e-3 / clones)
                tot["k_traverse"] += best
                depth += float(roots.path_len_tensor().float().mean()) - 1.0
                entries += sim + 1
            if fused is not None:
                ix_t, la_t = roots._ix, roots._la
                fused(actor.pool, ix_t, la_t, actor.pool[sim + 1], rew, val, pol)
                back = lambda c: c.backprop_tensors(sim + 1, rew, val, pol)
            else:
                r_log, v_log, p_log = eng.recurrent_heads(net_in, actor.pool[sim + 1])
                back = lambda c: c.backprop_nets_tensors(sim + 1, r_log, v_log, eng.V, -eng.support, p_log)
            if sampled and fused is not None:
                # the MFMA kernel: 8 launches on 8 different input / output buffers
                outs = [torch.empty_like(actor.pool[0]) for _ in range(clones)]
                torch.cuda.synchronize()
                g3 = timed_graph(lambda: [fused(actor.pool, ix_t, la_t, o, rew, val, pol) for o in outs])
                best = 1e9
                for _ in range(replays):
                    a, b = ev(), ev()
                    a.record(); g3.replay(); b.record()
                    torch.cuda.synchronize()
                    best = min(best, a.elapsed_time(b) * 1e-3 / clones)
                tot["k_mlp_recurrent"] += best
                del g3, outs
            if sampled:
                # each snapshot must expand entry sim+1 exactly once per replay: re-snapshot before every replay
                best = 1e9
                for _ in range(replays):
                    snaps2 = [c.clone() for c in snaps]
                    torch.cuda.synchronize()
                    g2 = timed_graph(lambda: [back(c) for c in snaps2])
                    a, b = ev(), ev()
                    a.record(); g2.replay(); b.record()
                    torch.cuda.synchronize()
                    best = min(best, a.elapsed_time(b) * 1e-3 / clones)
                    del g2, snaps2
                tot["k_backprop"] += best
                if fused is not None and sim < S - 2:
                    # the kernel the launch-per-phase search launches: backup of this simulation + descent of the next
                    best = 1e9
                    for _ in range(replays):
                        snaps2 = [c.clone() for c in snaps]
                        torch.cuda.synchronize()
                        g4 = timed_graph(lambda: [c.backprop_traverse_tensors(sim + 1, rew, val, pol) for c in snaps2])
                        a, b = ev(), ev()
                        a.record(); g4.replay(); b.record()
                        torch.cuda.synchronize()
                        best = min(best, a.elapsed_time(b) * 1e-3 / clones)
                        del g4, snaps2
                    tot["k_backprop_traverse"] += best
                launches += 1
                del g, snaps, ins
            back(roots)
    torch.cuda.synchronize()
    times = {k: v / launches for k, v in tot.items()} if launches else {}
    return times, search, launches * clones * replays, (depth / launches if launches else None), (entries / launches if launches else None)


def tail_timing(actor, reps=6):
    """Duration of everything a lock-step does after the search -- the two launches of include/hz_movetail.h -- on a LIVE position:
    one more search is run, what the tail reads and overwrites is saved (env states and generator positions, legal masks,
    uniforms, trajectory lengths, windows), and a hipGraph of `reps` x [restore, tail] is timed with HIP events; a graph of the
    restores alone is timed the same way and subtracted.  Every replica sees the same searched trees and the same games, so it
    picks the same (legal) actions and ends the same games as the product's lock-step would.  Returns seconds per tail and
    the share of envs whose game ended in it.  (Leaves the actor's games in disarray: call it last.)"""
    import torch
    if not actor.fused_tail:
        return None
    actor._search_part()
    torch.cuda.synchronize()
    env = actor.env
    snap = env.snapshot()
    saved = [(t, t.clone()) for t in (actor.legal, actor.uniform, actor.traj_len, actor.ent_sum, actor.move_count, actor.stack_buf,
                                      actor.out_count)]

    def graph(with_tail):
        side = torch.cuda.Stream(device=actor.device)
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(reps):
                for live, copy in saved:
                    live.copy_(copy)
                env.restore(snap)
                if with_tail:
                    actor._tail_part(True)
        return g
    out = []
    for with_tail in (True, False):
        g = graph(with_tail)
        best = 1e9
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e-3 / reps)
        out.append(best)
        if with_tail:
            ended = float(env.done.float().mean())
    return max(out[0] - out[1], 1e-9), ended


# ---------------------------------------------------------------------------------------------------- one self-play run
class Run:
    """One actor set-up + timed loop of `steps` lock-steps (the driver's contract for the primary run; the `also` runs reuse it)."""

    def __init__(self, args, workload, dtype_name, device, rank, world, net="random", rows_per_workgroup=0):
        import torch
        from hanabizero_amd.config import make_config
        from hanabizero_amd.selfplay import ActorGroup, SelfPlayActor
        self.args, self.rank, self.world, self.device = args, rank, world, device
        self.game, self.N, self.S, self.stack = WORKLOADS[workload]
        self.workload, self.dtype_name, self.net = workload, dtype_name, net
        self.flush_every = args.flush_every or (15 if self.game == "Hanabi-Small" else 40)
        self.cfg = make_config(self.game, simulations=self.S, stack=self.stack, p_mcts_num=self.N)
        # ("fp16x2": the fp32 engine whose recurrent inference is the MFMA kernel's fp16-pair build -- include/hz_mlp.h, HZ_F16X2)
        self.dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp16x2": torch.float32}[dtype_name]
        self.fused = "fp16x2" if dtype_name == "fp16x2" else (False if args.no_fused_mlp else None)
        self.engine = build_engine(self.cfg, self.dtype, device, fused=self.fused, net=net)
        K, N = args.actors_per_gpu, self.N
        assert N % K == 0
        # outbox ring: 8 x envs games -- the asynchronous drain hands a flush interval's games out one interval later, and
        # random-init play finishes a game every ~16 moves
        self.actors = [SelfPlayActor(self.cfg, self.engine, N // K, seed=0, device=device, use_graph=not args.no_graph,
                                     env_id_base=rank * N + k * (N // K), outbox_games=8 * (N // K),
                                     stream=torch.cuda.Stream(device=device) if K > 1 else None,
                                     predicted_lines={"auto": "auto", "on": True, "off": False}[args.predicted_lines]) for k in range(K)]
        for a in self.actors:
            a.mcts.rows_per_workgroup = rows_per_workgroup
        self.group = None
        if args.branch_graph and K > 1:
            for a in self.actors:
                a.stream = None
            self.group = ActorGroup(self.actors)
        self.games = self.rec_bytes = 0
        self.flush_s = self.wait_s = self.exchange_s = self.landing_s = 0.0
        self.env_ids = set()
        self._pending = False

    def step_all(self):
        if self.group is not None:
            self.group.step()
        else:
            for a in self.actors:
                a.step()

    def _land(self, a, packed):
        """One actor's packed finished games -> the replay owner (rank 0), on the actor's drain stream: a device-to-device
        gather (hanabizero_amd.dist.gather_packed) landing in pinned host memory, where `unpack_packed` views them."""
        import torch
        from hanabizero_amd.dist import gather_packed
        from hanabizero_amd.selfplay import packed_layout, unpack_packed
        from hanabizero_amd.dist import last_gather
        with torch.cuda.stream(a.drain_stream):
            got = gather_packed(packed, a.A, a.W, dst=0)
        self.exchange_s += last_gather["exchange_s"]  # (the all_gather of counts + the point-to-point receives: what this rank waited for)
        self.landing_s += last_gather["landing_s"]    # (the copies into pinned host memory)
        if self.rank == 0 and got:
            for buf, n, moves in got:
                self.games += n
                self.rec_bytes += packed_layout(n, moves, a.A, a.W)[1]
                if self.args.check_env_ids:
                    self.env_ids.update(int(x) for x in unpack_packed(buf, n, moves, a.A, a.W)["meta"][:, 2])

    def flush(self, final=False):
        """Finished games leave the actors WITHOUT stalling the lock-steps: every flush point finishes the drain whose
        snapshot was requested one interval earlier (the host waits for that snapshot only -- the GPU meanwhile has the
        interval's lock-steps queued) and requests the next snapshot; the final one waits for everything.
        --sync-drain: the round-1 behaviour (device-wide synchronize + blocking drain at every flush point)."""
        import torch
        t0 = time.perf_counter()
        if self.args.sync_drain:
            torch.cuda.synchronize()
            for a in self.actors:
                self._land(a, self._blocking(a))
        else:
            for a in self.actors:
                if self._pending:
                    tw = time.perf_counter()
                    a._snap.synchronize()  # the snapshot of one interval ago: the GPU has this interval's lock-steps queued meanwhile
                    self.wait_s += time.perf_counter() - tw
                    self._land(a, a.drain_end())
                if final:
                    self._land(a, self._blocking(a))
                else:
                    a.drain_begin()
            self._pending = not final
        self.flush_s += time.perf_counter() - t0

    @staticmethod
    def _blocking(a):
        a.drain_begin()
        return a.drain_end()

    def setup(self):
        import torch
        if not self.args.no_graph:  # capture (2 eager lock-steps + the capture itself) is set-up, whatever --warmup says
            if self.group is not None:
                self.group._capture() if self.group._graph is None else None
            else:
                for a in self.actors:
                    a._capture() if a._graph is None else None
            self.step_all()  # the first replay instantiates / uploads the graph (tens of ms): set-up as well
            torch.cuda.synchronize()

    def timed(self, steps, warmup, barrier):
        import torch
        import torch.distributed as dist
        for _ in range(warmup):
            self.step_all()
        self.flush(final=True)
        for _ in range(3):  # (an actor whose first drains changed its search kernels -- SelfPlayActor(predicted_lines="auto") -- captures
            self.step_all()  # its lock-step again at its next move: not inside the timed region)
        self.flush(final=True)
        if self.world > 1:  # the record gather's point-to-point channels exist before the timed region even if no game has ended yet
            w = torch.zeros(16, dtype=torch.uint8, device=self.device if self.args.backend == "nccl" else "cpu")
            dist.gather(w, [torch.empty_like(w) for _ in range(self.world)] if self.rank == 0 else None, dst=0)
        torch.cuda.synchronize()
        barrier()
        self.games, self.rec_bytes, self.flush_s, self.wait_s, self.exchange_s, self.landing_s = 0, 0, 0.0, 0.0, 0.0, 0.0
        self.env_ids = set()
        t0 = time.perf_counter()
        for k in range(steps):
            self.step_all()
            if (k + 1) % self.flush_every == 0 and k + 1 < steps:
                self.flush()
        self.flush(final=True)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.device if self.args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert all(int(a.illegal_steps) == 0 for a in self.actors), "an actor produced an illegal move"
        return elapsed

    def release(self):
        import torch
        self.actors, self.group, self.engine = [], None, None
        torch.cuda.empty_cache()


def net_error(game, dtype, search=True, fused=None):
    from tests.netgold import golden_net_error
    g = "Hanabi-Full" if game.startswith("Hanabi-Full") else game  # (the 5p net is the Full net at other widths)
    e = golden_net_error(g, dtype, fused=fused)
    w = e["wide"]
    out = {"against": "tests/golden/nets_%s.npz (reference MuZeroNet%s fp32 outputs)" % (g, "" if g == "Hanabi-Small" else "Full"),
           "path": ("fp32 GEMMs (root) + fused MFMA kernel, fp16 pairs (recurrent)" if fused == "fp16x2" else "fused MFMA kernels") if e["fused"] else "GEMM chain", "measure": "max / mean of |got - ref| / max(1, |ref|)",
           "worst": e["worst"], **{k: v for k, v in e.items() if isinstance(v, dict) and "max" in v},
           # the reference's own search precision as the yardstick (tests/golden/nets_*_autocast.npz)
           "reference_under_fp16_autocast_vs_its_fp32_worst": e["reference_autocast_vs_fp32"]["worst"],
           "vs_reference_under_fp16_autocast_worst": e["vs_reference_autocast"]["worst"],
           "rms_ratio_to_reference_autocast_256_rows": {k: v["rms"] / w["reference_autocast_vs_fp32"][k]["rms"] for k, v in w["got_vs_fp32"].items()}}
    if search and e["fused"] and fused != "fp16x2":
        from tests.netgold import search_divergence
        out["search_vs_fp32_engine"] = search_divergence(g, dtype, roots=512)
    return out


def _poll_giveups():
    from hanabizero_amd._lib import poll_giveups
    return poll_giveups()


def git_head():
    """The commit of this tree: from git where there is a checkout, else from the stamp __graft_entry__.build() leaves beside the
    library (the snapshot on a GPU box carries no .git)."""
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, "hanabizero_amd", "libhanabizero_hip.commit.txt")).read().strip() or None
    except Exception:
        return None


def spawn_ranks(n, script=None):
    """Launcher of `python bench.py --gpus N` (no torch.distributed.run around it; `script`: another entry point with the same
    contract, tools/loop_bench.py): N children of this same command line, rank r
    on GPU r, rendezvous on 127.0.0.1 at a free port; their stdout / stderr pass straight through (only rank 0 prints the JSON
    line).  Returns the worst exit code; if a rank dies the others are terminated rather than left at a barrier."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + sys.argv[1:], env=env))
    worst, live = 0, list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in live:  # (exactly the children started above)
                    q.terminate()
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="full4096", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp32", "fp16x2"],
                    help="format of the nets (weights, activations, hidden-state pool; fp32 accumulate): fp16 = the reference's own search "
                         "precision (autocast, core/mcts.py:38-40; the default), bf16 = what BASELINE.json configs[2] names (measured under `also`)")
    ap.add_argument("--predicted-lines", default="auto", choices=["auto", "on", "off"],
                    help="the search kernels' build (hz_search_set_predicted_lines): auto = the actor follows its searches' path lengths (the product)")
    ap.add_argument("--net", default="random", help='"random" (SURVEY 8d) or "sharp[:scale]" (concentrated policy: deep paths)')
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-fused-mlp", action="store_true", help="hipBLASLt GEMM chain instead of the fused MFMA kernel")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--flush-every", type=int, default=None, help="hand finished games to the replay owner every this many steps and once at the end of the timed loop (default 40 for Hanabi-Full, 15 for Hanabi-Small, whose games are shorter)")
    ap.add_argument("--sync-drain", action="store_true", help="round-1 drain: device-wide synchronize + blocking drain at every flush point")
    ap.add_argument("--check-env-ids", action="store_true", help="collect the global env ids of the gathered games (rehearsal tests)")
    ap.add_argument("--actors-per-gpu", type=int, default=1,
                    help="split this GPU's envs over this many concurrent actors (own hipGraph + stream each)")
    ap.add_argument("--branch-graph", action="store_true", help="with --actors-per-gpu > 1: one hipGraph with a branch per actor")
    ap.add_argument("--rows-per-workgroup", type=int, default=0, choices=[0, 16, -16, 32, -32], help="force a shape of the search kernel (include/hz_search.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra configurations measured after the primary run")
    ap.add_argument("--phase-kernels", action="store_true", help="roofline pass: also time the launch-per-phase kernels (`other`)")
    ap.add_argument("--cpu-sample-trees", type=int, default=4096)
    ap.add_argument("--cpu-sample-moves", type=int, default=16)
    ap.add_argument("--cpu-procs", type=int, default=16, help="upper bound on the processes of the all-cores CPU leg (a 1-GPU box's CPU share)")
    ap.add_argument("--cpu-worker", default=None, choices=["both", "tree", "env", "ref_tree", "ref_env"], help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker:  # a child of cpu_baseline(): oracle only, before anything imports torch
        game, _, S, _ = WORKLOADS[args.workload]
        A = {"Hanabi-Small": 11, "Hanabi-Full": 20, "Hanabi-Full-5p": 48}[game]
        v, dt = cpu_worker(args.cpu_worker, game, A, S, args.cpu_sample_trees, args.cpu_sample_moves)
        print(json.dumps({"moves_per_s": v, "seconds": dt}), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this process becomes the launcher -- it starts one rank per GPU (as the reference
        # starts its own workers, core/train.py:463-474) BEFORE anything here has imported torch or touched a GPU, relays what
        # they print (rank 0's JSON line) and leaves with the worst exit code.  Under torch.distributed.run WORLD_SIZE is set
        # and this branch is not taken.
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or with no WORLD_SIZE in the environment (bench.py then starts its own ranks)" % (args.gpus, world, args.gpus)
    if args.share_device:
        assert args.backend == "gloo", "--share-device is a gloo rehearsal mode"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    from hanabizero_amd.dist import reserve_landing
    run = Run(args, args.workload, args.dtype, device, rank, world, net=args.net, rows_per_workgroup=args.rows_per_workgroup)
    game, N, S, stack = run.game, run.N, run.S, run.stack
    if rank == 0:  # the replay owner's pinned landing buffers (about 4.5 KB per finished Hanabi-Full game)
        reserve_landing(16384 * N, world)
    run.setup()
    elapsed = run.timed(args.steps, args.warmup, barrier)

    moves = world * N * args.steps
    out = {
        "metric": "selfplay_moves_per_sec",
        "value": moves / elapsed,
        "unit": "moves/s",
        "sims_per_sec": moves * (S - 1) / elapsed,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 tree + integer env (bit-exact), %s nets" % args.dtype,
        "data": "synthetic",
        "config": {"workload": "%s, %d envs/GPU, %d sims/move (%d run, as the reference), stack %d, %s nets, global obs" % (
                       game + ("" if game.endswith("5p") else " 2p"), N, S, S - 1, stack,
                       "random-init (heads N(0,0.1))" if args.net == "random" else "random-init with a sharpened policy head (%s)" % args.net),
                   "envs_per_gpu": N, "actors_per_gpu": args.actors_per_gpu, "simulations": S, "hipgraph": not args.no_graph,
                   "parallelism": "actor-per-GPU x%d" % world, "drain": "synchronous" if args.sync_drain else "asynchronous (side stream, one interval behind)",
                   "games_finished": run.games, "record_bytes_gathered": run.rec_bytes,
                   # host time of the flush points: the drain work proper (pack + gather + landing copy; under --sync-drain also the
                   # device-wide synchronize in front of it, which is what stalls the actors) and, asynchronous mode only, the
                   # host's wait for the snapshot of one interval ago -- the GPU runs the queued lock-steps meanwhile
                   "drain_gather_ms_total": 1e3 * (run.flush_s - run.wait_s), "drain_snapshot_wait_ms_total": 1e3 * run.wait_s,
                   # of drain_gather_ms_total, on rank 0: waiting in the exchange (counts all_gather + receives from the other ranks:
                   # 0 at N = 1) and copying the received buffers into pinned host memory
                   "gather_exchange_ms_total": 1e3 * run.exchange_s, "gather_landing_ms_total": 1e3 * run.landing_s},
    }
    if args.check_env_ids and rank == 0:
        out["config"]["env_id_min_max_distinct"] = [min(run.env_ids), max(run.env_ids), len(run.env_ids)] if run.env_ids else None
    if rank == 0 and not args.no_roofline:
        out["net_error"] = net_error(game, run.dtype, fused=run.fused if run.fused == "fp16x2" else None)

    if rank == 0 and not args.no_roofline:
        actor, engine, cfg, dtype = run.actors[0], run.engine, run.cfg, run.dtype
        times, search, launches, dbar, sbar = kernel_timing(actor, sample_sims=(4, 16, 28, 40) if (args.phase_kernels or engine.fused is None) else ())
        if dbar is None:
            dbar, _ = mean_path_edges(actor)
            sbar = (S - 1) / 2.0
        A, H, e = cfg.action_space_size, engine.H, (4 if dtype == torch.float32 else 2)
        V = engine.V
        Nk = actor.N  # trees per launch (per actor)
        # algorithmic bytes per launch (SURVEY.md 8d per-tree figures x N trees per launch)
        b_trav = Nk * (16 * A * dbar + (0 if engine.fused is not None else 2 * H * e))
        fused_on = engine.fused is not None
        b_back = Nk * ((4 * A + 16 * A + 16 + 8 + 16 * (dbar + 1) + 12 * sbar) if fused_on else
                       (A * e + 2 * V * e + 16 * A + 16 + 16 * (dbar + 1) + 12 * sbar))
        other = {}
        if times:
            kern = {"k_traverse": (b_trav, times["k_traverse"]), "k_backprop": (b_back, times["k_backprop"])}
            if fused_on and times.get("k_backprop_traverse", 0) > 0:
                kern["k_backprop_traverse"] = (b_trav + b_back, times["k_backprop_traverse"])
            other = {k: {"bound": "hbm", "avg_launch_us": t * 1e6, "bytes_per_launch": b, "GBps": b / t / 1e9, "frac": b / t / 1e9 / HBM_PEAK_GBS}
                     for k, (b, t) in kern.items()}
        flops = engine.flops_per_sample() * Nk
        if fused_on and times:
            t = times["k_mlp_recurrent"]
            other["k_mlp_recurrent"] = {"bound": "mfma", "avg_launch_us": t * 1e6, "flop_per_launch": flops,
                                        "TFLOPps": flops / t / 1e12, "frac": flops / t / 1e12 / MFMA_PEAK_TFLOPS,
                                        "weight_bytes_per_wg": engine.fused.weight_bytes_per_wg}
        # the env / actor kernels of a lock-step: everything after the search = the two launches of include/hz_movetail.h.
        # Algorithmic HBM bytes per env and move: the game's state line in and out (2 x 128 B), <= 8 generator
        # words, the root's child records (16 A) and sums, legal masks in / out, counts / values / action out, the history's rows
        # (visits 2 A, observation 4 W twice -- after the move and, identical unless the game ended, as the next head --, legal A
        # twice, reward / action / value, meta 16), the next move's noise (4 A) and uniform, the model's input window moving up
        # one slot ((2 stack - 1) slots of Dp elements), and for the share of envs whose game ended the seven trajectory rows
        # out of the history into the outbox (read + write) plus the new game's generator words
        tt = tail_timing(actor) if engine.fused is not None else None
        if tt is not None:
            t_tail, ended = tt
            Wp, T, st_n, es = actor.W, actor.T, actor.stack, actor.stack_buf.element_size()
            rows = T + T + 4 * T + 2 * T * A + (T + 1) * A + 4 * (T + 1) * Wp + 16
            per_env = (2 * 128 + 32 + 16 * A + 8 + A + 8 + 4 * A + 4 + 4 + 2 * A + 2 * 4 * Wp + 2 * A + 6 + 16 + 4 * A + 8 + 16 + 16 +
                       (2 * st_n - 1) * actor.Dp * es + ended * (2 * rows + 4 * 2 * 25))
            btail = Nk * per_env
            other["move_tail"] = {"bound": "hbm", "kernels": "k_move_tail_a + k_move_tail_b (one wave per env: read-out, action, env step, "
                                  "history | finished games out, reset, observation, window; include/hz_movetail.h)",
                                  "avg_launch_us": t_tail * 1e6, "bytes_per_launch": btail, "GBps": btail / t_tail / 1e9,
                                  "frac": btail / t_tail / 1e9 / HBM_PEAK_GBS, "games_ended_share": ended,
                                  "note": "both launches together, timed on a live position (restore + tail replayed in a hipGraph, the "
                                          "restores' own time subtracted); latency-bound: ~15 dependent steps per env, not bandwidth"}
        traffic_all, traffic_src = {}, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                data = json.load(open(pmc))
                traffic_all = data.get(args.workload, {})
                traffic_src = {"file": "profiles/pmc_traffic.json", "taken_at_commit": traffic_all.get("_commit"),
                               "this_run_commit": git_head(), "note": "a committed measurement (two rocprofv3 --pmc passes, tools/pmc_traffic.py), not collected in this run"}
            except Exception:
                traffic_all = {}
        if fused_on and search is not None:
            # the persistent search kernel: all S-1 simulations of a move; MFMA work = (S-1) recurrent inferences, its tree phases
            # add the algorithmic HBM bytes of (S-1) backups and descents; every workgroup streams the whole weight set from L2
            # once per simulation
            t = search["mean_s"]
            fl = flops * (S - 1)
            rows_wg = 16 if (abs(actor.mcts.rows_per_workgroup) == 16 or (actor.mcts.rows_per_workgroup == 0 and (Nk + 15) // 16 <= torch.cuda.get_device_properties(device).multi_processor_count)) else 32
            wgs = (Nk + rows_wg - 1) // rows_wg
            l2_bytes = wgs * engine.fused.weight_bytes_per_wg * (S - 1) + 2 * Nk * H * e * (S - 1)
            pairs = run.fused == "fp16x2"  # (the fp16-pair build: the same algorithmic flops through three MFMAs per product, twice the stream)
            out["roofline"] = {"bound": "mfma", "kernel": "k_search_pairs" if pairs else "k_search", "achieved": fl / t / 1e12, "peak": MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": fl / t / 1e12 / MFMA_PEAK_TFLOPS,
                               "traffic": None if pairs else traffic_all.get("k_search"), "traffic_source": None if pairs else traffic_src,
                               "avg_launch_us": t * 1e6, "min_launch_us": search["min_s"] * 1e6, "launches_timed": search["launches"], "empty_event_pair_us": search["empty_event_pair_us"], "eager_step_us": search["eager_step_us"],
                               "poll_giveups": _poll_giveups(),  # waits on arrival counters that timed out in this process (must be 0)
                               "flop_per_launch": fl, "simulations_per_launch": S - 1, "trees_per_workgroup": rows_wg, "workgroups": wgs,
                               "predicted_line_kernels": bool(actor._lines_on),  # (SelfPlayActor's choice for this policy: hz_search_set_predicted_lines)
                               "l2_stream": {"bytes_per_launch": l2_bytes, "achieved_TBps": l2_bytes / t / 1e12, "peak_TBps": L2_PEAK_TBS,
                                             "frac": l2_bytes / t / 1e12 / L2_PEAK_TBS,
                                             "note": "L1<-L2 weight stream (every workgroup pulls all %d weight bytes per simulation) + pool rows: the resource that binds this kernel" % engine.fused.weight_bytes_per_wg},
                               "hbm_algorithmic": {"bytes_per_launch": (b_trav + b_back + 2 * Nk * H * e) * (S - 1),
                                                   "GBps": (b_trav + b_back + 2 * Nk * H * e) * (S - 1) / t / 1e9,
                                                   "frac_of_hbm_peak": (b_trav + b_back + 2 * Nk * H * e) * (S - 1) / t / 1e9 / HBM_PEAK_GBS},
                               "mean_path_edges": dbar, "mean_expanded_entries": sbar,
                               "method": "a HIP event on either side of the search launch in 32 further lock-steps of the live actor, enqueued kernel by kernel on the launch stream (search_in_step): mean (avg_launch_us) and shortest (min_launch_us); eager_step_us = those lock-steps end to end, to be read against ms_per_step",
                               "other": other}
        elif other:
            dom = max(other, key=lambda k: other[k]["avg_launch_us"])
            d = other[dom]
            out["roofline"] = {"bound": d["bound"], "kernel": dom, "achieved": d.get("GBps", d.get("TFLOPps")),
                               "peak": HBM_PEAK_GBS if d["bound"] == "hbm" else MFMA_PEAK_TFLOPS,
                               "unit": "GB/s" if d["bound"] == "hbm" else "TFLOP/s", "frac": d["frac"], "traffic": traffic_all.get(dom),
                               "traffic_source": traffic_src, "avg_launch_us": d["avg_launch_us"], "launches_timed": launches,
                               "mean_path_edges": dbar, "mean_expanded_entries": sbar, "other": other}

    if rank == 0 and world == 1 and not args.no_also:
        # other configurations, same invocation, same box (each: own actors + hipGraph; steps as the primary run)
        also = {}
        run.release()
        del run
        plan = []
        if args.workload == "full4096" and args.net == "random":
            other_dt = "bf16" if args.dtype != "bf16" else "fp16"
            plan = [("full8192", "full8192", args.dtype, "random"), ("full8192_bf16", "full8192", "bf16", "random"),  # configs[2] names bf16
                    (other_dt, args.workload, other_dt, "random"), ("deep_paths", args.workload, args.dtype, "sharp"),
                    ("deep_paths_full8192", "full8192", args.dtype, "sharp"),
                    # the engines inside the contract's 1e-3 of the reference's fp32 nets: hand-written recurrent kernel / library GEMMs
                    ("fp16x2", args.workload, "fp16x2", "random"), ("fp32", args.workload, "fp32", "random"),
                    ("small4096", "small4096", args.dtype, "random"),   # BASELINE configs[1]'s game at 4096 envs
                    ("full5p2048", "full5p2048", args.dtype, "random")]   # configs[4]'s game (A = 48), the self-play part alone
            plan = [p for k, p in enumerate(plan) if p[1:] not in [q[1:] for q in plan[:k]]]
        for name, wl, dt, net in plan:
            r = Run(args, wl, dt, device, 0, 1, net=net)
            r.setup()
            el = r.timed(args.steps, args.warmup, lambda: None)
            entry = {"workload": wl, "dtype": dt, "net": net, "value": r.N * args.steps / el, "unit": "moves/s",
                     "ms_per_step": 1e3 * el / args.steps, "steps": args.steps, "games_finished": r.games,
                     "predicted_line_kernels": bool(r.actors[0]._lines_on)}
            if not args.no_roofline:
                _, search, _, _, _ = kernel_timing(r.actors[0], sample_sims=())
                if search:
                    entry["k_search_avg_launch_us"], entry["k_search_min_launch_us"] = search["mean_s"] * 1e6, search["min_s"] * 1e6
                if net != "random":
                    entry["mean_path_edges"], entry["deepest_path_edges"] = mean_path_edges(r.actors[0])
                if dt != args.dtype:
                    ne = net_error(r.game, r.dtype, fused=r.fused if r.fused == "fp16x2" else None)
                    entry["net_error_worst"], entry["search_vs_fp32_engine"] = ne["worst"], ne.get("search_vs_fp32_engine")
            also[name] = entry
            r.release()
            del r
        # BASELINE configs[4] (Hanabi-Full 5p: self-play + reanalyze + learner batch 256) as its own process on this GPU, after
        # everything above has let go of it: tools/loop_bench.py's JSON line, trimmed (failures are reported, not raised)
        if args.workload == "full4096" and args.net == "random":
            try:
                torch.cuda.empty_cache()
                p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "loop_bench.py"), "--rounds", "3"], stdout=subprocess.PIPE,
                                   stderr=subprocess.PIPE, text=True, timeout=300)
                line = [l for l in p.stdout.splitlines() if l.startswith("{")]
                if p.returncode == 0 and line:
                    d = json.loads(line[-1])
                    also["config5_loop"] = {k: d[k] for k in ("workload", "learner_steps_per_s", "selfplay_moves_per_s", "replay_ratio_target", "replay_ratio_achieved",
                                                              "learner_steps", "wall_s", "weight_handover_ms", "host_ms_per_learner_step_enqueue",
                                                              "host_wait_for_the_gpu_ms_per_learner_step", "replay_positions", "learner_blocks", "reference")}
                else:
                    also["config5_loop"] = {"error": (p.stderr or p.stdout)[-400:]}
            except Exception as e:  # noqa: BLE001
                also["config5_loop"] = {"error": repr(e)[:400]}
        out["also"] = also
        # the other 16-bit format's figure for the SAME workload, at the top level beside `value` (BASELINE's configs name bf16, the
        # reference's own search runs in fp16: both are measured in every default run)
        for name, entry in also.items():
            if entry.get("workload") == args.workload and entry.get("net") == args.net and entry.get("dtype") not in (None, args.dtype):
                out["value_" + entry["dtype"]] = entry["value"]

    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (rank 0 at N = 1 only)
        cfgA = {"Hanabi-Small": 11, "Hanabi-Full": 20, "Hanabi-Full-5p": 48}[game]
        out["cpu_baseline"] = cpu_baseline(game, cfgA, S, args.cpu_sample_trees, args.cpu_sample_moves, args.workload, args.cpu_procs)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()  # rank 0 spent extra seconds on the roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
