#!/usr/bin/env python3
"""bench.py -- self-play moves/sec & MCTS sims/sec of the MI355X engine on BASELINE.json's headline workload.

One "step" = one lock-step of the self-play hot path over every env of this GPU: root inference -> root prepare ->
49 x (HIP select + gather -> dynamics/prediction GEMMs -> HIP expand/backup) -> read-out -> action sampling ->
HIP env step + encode -> finished-game flush -> reset.  Inputs are resident in HBM; nothing is skipped.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Besides the driver's contract fields it carries
  roofline      the dominant hand-written kernel: algorithmic bytes per launch / average launch duration measured here
                with HIP events on the launch stream (an instrumented eager pass over real search states after the
                timed region), against the 8 TB/s HBM peak; `traffic` from profiles/ PMC summaries when present
  cpu_baseline  the plain-C oracle (tree + env, no nets: the part the reference runs on CPU) timed on one host core
                on a bounded sample of the same workload.  A reported baseline, not the target.
"""
import argparse
import json
import os
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
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # same guide: ~2.5 PFLOP/s dense bf16


def build_engine(cfg, dtype, device, fused=None):
    import torch
    from hanabizero_amd.model import InferenceEngine
    torch.manual_seed(0)
    net = cfg.get_uniform_network()
    with torch.no_grad():  # fixed random init, zero-initialised heads perturbed with N(0, 0.1)
        for head in (net._prediction_value, net._dynamics_reward, net._prediction_actor):
            head[-1].weight.normal_(0, 0.1)
            head[-1].bias.normal_(0, 0.1)
    net.eval()
    return InferenceEngine(net, cfg.value_support.max, dtype=dtype, device=device, fused=fused)


def kernel_timing(actor, sample_sims=(4, 16, 28, 40), clones=8, replays=5):
    """Average launch duration of the two hand-written tree kernels on LIVE search states, with HIP events.

    An eager search of the actor's current position is run; at each sampled simulation the tree state is snapshotted
    into `clones` independent handles (hz_tree_copy) and a hipGraph of `clones` back-to-back launches -- one per
    snapshot, so no launch sees data the previous one left in cache -- is replayed between two HIP events on the
    launch stream (torch's current stream).  An eager launch cannot be timed this way: on this stack an empty event
    pair already reads ~26 us.  The figure includes the ~1-2 us dependent-launch boundary between graph nodes, which
    rocprofv3's per-dispatch durations (profiles/) exclude.
    Returns {kernel: (avg seconds per launch, launches)}, mean path edges, mean expanded entries."""
    import torch
    cfg, roots, eng = actor.cfg, actor.roots, actor.engine
    N, S, oh = actor.N, actor.S, eng.onehot_cols
    actor._draw()
    value0, logits0, hidden0 = actor.root_inference()
    roots.prepare(cfg.root_exploration_fraction, actor.noise, actor.zeros_n, logits0, actor.legal)
    actor.pool[0].copy_(hidden0)
    net_in = torch.empty((N, eng.H + oh), dtype=eng.dtype, device=actor.device)
    fused = getattr(eng, "fused", None)
    tot = {"k_traverse": 0.0, "k_backprop": 0.0, "k_mlp_recurrent": 0.0, "k_backprop_traverse": 0.0}
    rew = torch.empty(N, dtype=torch.float32, device=actor.device)
    val = torch.empty(N, dtype=torch.float32, device=actor.device)
    pol = torch.empty((N, actor.A), dtype=torch.float32, device=actor.device)
    launches, depth, entries = 0, 0.0, 0.0
    ev = lambda: torch.cuda.Event(enable_timing=True)
    t_search = None

    def timed_graph(body):
        side = torch.cuda.Stream(device=actor.device)
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            body()
        return g

    fused16 = eng.fused_shape(16, 2) if (fused is not None and getattr(actor.mcts, "persistent", False)) else None
    if fused16 is not None:
        # the kernel the product path launches: ONE persistent search kernel per move.  4 launches per graph, each on
        # its own snapshot of the freshly prepared trees (a finished tree cannot be searched again), best of 3 replays
        best = 1e9
        for _ in range(3):
            snaps = [roots.clone() for _ in range(4)]
            torch.cuda.synchronize()
            g0 = timed_graph(lambda: [c.search_tensors(fused16, actor.pool, S - 1, rew, val, pol) for c in snaps])
            a, b = ev(), ev()
            a.record(); g0.replay(); b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e-3 / 4)
            del g0, snaps
        t_search = best
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
                # the kernel the fused search actually launches: backup of this simulation + descent of the next
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
    times = {k: v / launches for k, v in tot.items()}
    if t_search is not None:
        times["k_search"] = t_search
    return times, launches * clones * replays, depth / launches, entries / launches


def cpu_baseline(game, A, S, sample_trees, moves):
    """Oracle tree + oracle env on ONE host core: prepare + (S-1) x (traverse, backprop with recorded fake-net outputs)
    + env step + encode per move.  Returns moves/s."""
    import numpy as np
    from oracle.cport import OracleEnv, OracleTree
    rng = np.random.RandomState(0)
    N = sample_trees
    env = OracleEnv(game, np.arange(N))
    env.reset()
    obs, legal = env.observe()
    tree = OracleTree(N, A, S, seed=0, value_delta_max=0.006)
    noises = rng.dirichlet([0.3] * A, N).astype(np.float32)
    logits0 = rng.randn(N, A).astype(np.float32)
    rew = (rng.randint(-1, 2, (S - 1, N)) * (rng.rand(S - 1, N) < 0.3)).astype(np.float32)
    val = (rng.rand(S - 1, N) * 25).astype(np.float32)
    lg = rng.randn(S - 1, N, A).astype(np.float32)
    zeros = np.zeros(N, np.float32)
    t0 = time.perf_counter()
    done_moves = 0
    for _ in range(moves):
        tree = OracleTree(N, A, S, seed=0, value_delta_max=0.006)  # the reference builds a new Roots per move
        tree.prepare(0.25, noises, zeros, logits0, legal)
        for sim in range(S - 1):
            tree.traverse(sim, 19652, 1.25, 0.999)
            tree.backprop(sim + 1, 0.999, rew[sim], val[sim], lg[sim])
        dist = tree.distributions().astype(np.float64) * legal
        act = (dist + 1e-3 * legal).argmax(1).astype(np.int32)
        _, done, _ = env.step(act)
        if done.any():
            env.reset(done)
        obs, legal = env.observe()
        done_moves += N
    dt = time.perf_counter() - t0
    return done_moves / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="full4096", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--no-fused-mlp", action="store_true", help="hipBLASLt GEMM chain instead of the fused MFMA kernel")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--flush-every", type=int, default=None, help="drain + gather finished games every this many steps and once at the end of the timed loop (default 40 for Hanabi-Full -- the outbox ring holds 4 x envs games: ~58 steps of random-init play -- and 15 for Hanabi-Small, whose games are shorter)")
    ap.add_argument("--actors-per-gpu", type=int, default=1,
                    help="split this GPU's envs over this many concurrent actors (own hipGraph + stream each)")
    ap.add_argument("--branch-graph", action="store_true", help="with --actors-per-gpu > 1: one hipGraph with a branch per actor")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample-trees", type=int, default=4096)
    ap.add_argument("--cpu-sample-moves", type=int, default=24)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
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

    from hanabizero_amd.config import make_config
    from hanabizero_amd.dist import gather_packed, reserve_landing
    from hanabizero_amd.selfplay import SelfPlayActor, packed_layout

    game, N, S, stack = WORKLOADS[args.workload]
    if args.flush_every is None:
        args.flush_every = 15 if game == "Hanabi-Small" else 40
    cfg = make_config(game, simulations=S, stack=stack, p_mcts_num=N)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    engine = build_engine(cfg, dtype, device, fused=False if args.no_fused_mlp else None)
    K = args.actors_per_gpu
    assert N % K == 0
    actors = [SelfPlayActor(cfg, engine, N // K, seed=0, device=device, use_graph=not args.no_graph,
                            env_id_base=rank * N + k * (N // K),
                            stream=torch.cuda.Stream(device=device) if K > 1 else None) for k in range(K)]
    actor = actors[0]
    group = None
    if args.branch_graph and K > 1:
        from hanabizero_amd.selfplay import ActorGroup
        for a in actors:
            a.stream = None
        group = ActorGroup(actors)

    def step_all():
        if group is not None:
            group.step()
        else:
            for a in actors:
                a.step()

    def barrier():
        if world > 1:
            dist.barrier()

    games, rec_bytes = 0, 0
    if rank == 0:  # the replay owner's pinned landing buffers (about 4.5 KB per finished Hanabi-Full game)
        reserve_landing(16384 * N, world)

    flush_s = 0.0

    def flush():
        """Finished games -> the replay owner (rank 0): one packed byte buffer per actor, gathered device to device
        (hanabizero_amd.dist.gather_packed), landing in pinned host memory on rank 0 where `unpack_packed` views it."""
        nonlocal games, rec_bytes, flush_s
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for a in actors:
            got = gather_packed(a.drain_packed(), a.A, a.W, dst=0)
            if rank == 0 and got:
                for buf, n, moves in got:
                    games += n
                    rec_bytes += packed_layout(n, moves, a.A, a.W)[1]
        flush_s += time.perf_counter() - tf0

    if not args.no_graph:  # capture (2 eager lock-steps + the capture itself) is set-up, whatever --warmup says
        if group is not None:
            group._capture() if group._graph is None else None
        else:
            for a in actors:
                a._capture() if a._graph is None else None
        step_all()  # the first replay instantiates / uploads the graph (tens of ms): set-up as well
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step_all()
    flush()
    if world > 1:  # the record gather's point-to-point channels exist before the timed region even if no game has ended yet
        w = torch.zeros(16, dtype=torch.uint8, device=device if args.backend == "nccl" else "cpu")
        dist.gather(w, [torch.empty_like(w) for _ in range(world)] if rank == 0 else None, dst=0)
    torch.cuda.synchronize()
    barrier()
    games, rec_bytes, flush_s = 0, 0, 0.0
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_all()
        if (k + 1) % args.flush_every == 0:
            flush()
    flush()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert all(int(a.illegal_steps) == 0 for a in actors), "an actor produced an illegal move"

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
        "config": {"workload": "%s 2p, %d envs/GPU, %d sims/move (%d run, as the reference), stack %d, random-init "
                               "nets (heads N(0,0.1)), global obs" % (game, N, S, S - 1, stack),
                   "envs_per_gpu": N, "actors_per_gpu": K, "simulations": S, "hipgraph": not args.no_graph, "parallelism": "actor-per-GPU x%d" % world,
                   "games_finished": games, "record_bytes_gathered": rec_bytes,
                   "drain_gather_ms_total": 1e3 * flush_s},
    }

    if rank == 0 and not args.no_roofline:
        times, launches, dbar, sbar = kernel_timing(actor)
        A, H, e = cfg.action_space_size, engine.H, (4 if dtype == torch.float32 else 2)
        V = engine.V
        # algorithmic bytes per launch (SURVEY.md 8d per-tree figures x N trees per launch)
        Nk = actor.N  # trees per launch (per actor)
        b_trav = Nk * (16 * A * dbar + (0 if engine.fused is not None else 2 * H * e))                          # child rows per level + hidden row in and out
        # fused backup: policy logits + the two categorical head rows in (net dtype), child rows out, header, backup, min-max
        b_back = Nk * (A * e + 2 * V * e + 16 * A + 16 + 16 * (dbar + 1) + 12 * sbar)
        fused_on = engine.fused is not None
        if fused_on:  # plain backup: fp32 reward/value/policy in (SURVEY 8d formula)
            b_back = Nk * (4 * A + 16 * A + 16 + 8 + 16 * (dbar + 1) + 12 * sbar)
        kern = {"k_traverse": (b_trav, times["k_traverse"]), "k_backprop": (b_back, times["k_backprop"])}
        if fused_on and times.get("k_backprop_traverse", 0) > 0:
            kern["k_backprop_traverse"] = (b_trav + b_back, times["k_backprop_traverse"])
        other = {k: {"bound": "hbm", "avg_launch_us": t * 1e6, "bytes_per_launch": b, "GBps": b / t / 1e9, "frac": b / t / 1e9 / HBM_PEAK_GBS}
                 for k, (b, t) in kern.items()}
        flops = engine.flops_per_sample() * Nk
        if fused_on:
            t = times["k_mlp_recurrent"]
            other["k_mlp_recurrent"] = {"bound": "mfma", "avg_launch_us": t * 1e6, "flop_per_launch": flops,
                                        "TFLOPps": flops / t / 1e12, "frac": flops / t / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                        "weight_bytes_per_wg": engine.fused.weight_bytes_per_wg}
        if fused_on and "k_search" in times:
            # the persistent search kernel: all S-1 simulations of a move; MFMA work = (S-1) recurrent inferences, its
            # tree phases add the algorithmic HBM bytes of (S-1) backups and descents
            t = times["k_search"]
            fl = flops * (S - 1)
            other["k_search"] = {"bound": "mfma", "avg_launch_us": t * 1e6, "flop_per_launch": fl,
                                 "TFLOPps": fl / t / 1e12, "frac": fl / t / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                 "simulations_per_launch": S - 1,
                                 "weight_bytes_streamed_per_cu_per_simulation": engine.fused.weight_bytes_per_wg,
                                 "tree_bytes_per_launch": (b_trav + b_back) * (S - 1)}
        traffic_all = {}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic_all = json.load(open(pmc)).get(args.workload, {})
            except Exception:
                traffic_all = {}
        dom = max(other, key=lambda k: other[k]["avg_launch_us"])
        d = other[dom]
        if d["bound"] == "hbm":
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": d["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": d["frac"], "traffic": traffic_all.get(dom)}
        else:
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": d["TFLOPps"], "peak": MFMA_BF16_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": d["frac"], "traffic": traffic_all.get(dom)}
        out["roofline"].update({"avg_launch_us": d["avg_launch_us"], "launches_timed": launches,
                                "mean_path_edges": dbar, "mean_expanded_entries": sbar,
                                "method": "HIP events around hipGraph replays of back-to-back launches on independent snapshots of live search states (4 for k_search, 8 for the per-phase kernels)",
                                "other": other})

    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (rank 0 at N = 1 only)
        v, dt = cpu_baseline(game, cfg.action_space_size, S, args.cpu_sample_trees, args.cpu_sample_moves)
        out["cpu_baseline"] = {"value": v, "unit": "moves/s", "cores": 1, "kind": "port",
                               "sample": "oracle tree+env (no nets), %d envs x %d moves x %d sims, %.1f s on 1 core" % (
                                   args.cpu_sample_trees, args.cpu_sample_moves, S - 1, dt)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()  # rank 0 spent extra seconds on the roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
