#!/usr/bin/env python3
"""tools/loop_bench.py -- BASELINE.json configs[4] timed as ONE loop per rank, one JSON line: Hanabi-Full 5 players (A = 48,
D = 1385, mdp global), 50 simulations per move, self-play + reanalyze + learner batch 256, on N GPUs of one node.

Role map (what the reference runs as Ray actors: /root/reference/core/train.py:317-431, 440-481,
core/reanalyze_worker.py:45-86, 249-304, 307-440, core/selfplay_worker.py:91-393 -- here one loop per rank, no control plane):

  rank 0            the LEARNER rank: owns the replay (hanabizero_amd.device_replay.DeviceReplay: the actors' packed records,
                    bit-packed, in HBM), the target model (reanalyze searches + value targets), the learner's module and its
                    captured step (learner.LearnerPipeline: batch k + 1 is sampled, re-searched and assembled on one stream
                    while step k trains on another; nothing on the host but the enqueueing).  At N = 1 it is also the actor.
  ranks 1 .. N - 1  ACTORS: one SelfPlayActor each under its hipGraph (env ids keyed by rank: no two ranks play the same game).
  a round           every actor plays `--flush-every` lock-steps; its finished games go to rank 0 as ONE packed buffer that never
                    leaves the devices (dist.gather_packed(to_host=False): RCCL point-to-point into reused receive buffers) and
                    are appended to the replay on the device; rank 0 then trains `ratio x moves of the round` steps (the
                    reference's replay ratio, README.md:51: 0.008 learner steps per self-play move); the round ends with a
                    two-number broadcast (trained steps, "weights follow") and, every checkpoint_interval steps, the weights
                    (dist.broadcast_weights: one flat buffer per dtype) which every actor takes over IN PLACE on its device
                    (InferenceEngine.load from a device-resident module: no host copy, the captured lock-step sees them).

Why actors wait for the learner: one MI355X plays 1.1 M Hanabi-Full-5p moves/s; at ratio 0.008 that is owed 9 k learner steps/s,
and a learner step (256 x 6 inferences forward and backward, ~1.7 k small kernels) takes ~8 ms.  At the reference's ratio the
configuration is LEARNER-bound by two orders of magnitude on any number of GPUs -- the actors of a round are done in
`flush_every x 2 ms` and wait; `--ratio` sets another exchange rate, `--ratio 0` free-runs the actors (self-play + ingest only).
Reported: learner steps/s, self-play moves/s of the whole job, the ratio achieved, the host's share (time spent enqueueing),
one weight hand-over timed on its own.

Launch: `python tools/loop_bench.py --gpus N` starts its own N ranks (as bench.py does); under torch.distributed.run it reads
RANK / LOCAL_RANK / WORLD_SIZE.  `--backend gloo --share-device` rehearses N ranks on one GPU."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--game", default="Hanabi-Full-5p")
    ap.add_argument("--envs", type=int, default=2048, help="per acting rank")
    ap.add_argument("--rounds", type=int, default=4, help="timed rounds of --flush-every lock-steps each")
    ap.add_argument("--warm-rounds", type=int, default=6, help="self-play only, to fill the replay (the reference waits for start_window_size positions, train.py:352-359)")
    ap.add_argument("--flush-every", type=int, default=10)
    ap.add_argument("--ratio", type=float, default=0.008, help="learner steps per self-play move (reference README.md:51); 0: self-play + ingest only")
    ap.add_argument("--max-steps-per-round", type=int, default=0, help="cap on the learner steps of one round (0: none)")
    ap.add_argument("--reanalyze-share", type=float, default=0.5, help="share of every batch whose policy targets are re-searched")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--checkpoint-interval", type=int, default=0, help="learner steps between weight hand-overs to the actors (0: the config's)")
    ap.add_argument("--target-interval", type=int, default=0, help="learner steps between target-model refreshes (0: the config's, 200)")
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--simulations", type=int, default=50)
    ap.add_argument("--replay-capacity", type=int, default=4_000_000, help="positions the device replay holds (Hanabi-Full 5p: ~330 B each)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--learner-acts", action="store_true", help="N > 1: rank 0 plays too (default: it only learns)")
    ap.add_argument("--actor-stream", default="null", choices=["null", "normal", "low"], help="the stream the lock-steps run on")
    ap.add_argument("--parallel-heads", type=int, default=0, help="head chains of the learner's inferences on streams of their own (FusedTrainNet)")
    ap.add_argument("--research-streams", type=int, default=2, help="streams the batches' re-searches alternate on (0: on the prepare stream, two batch slots)")
    ap.add_argument("--slots", type=int, default=0, help="batches in flight (0: the pipeline's default)")
    ap.add_argument("--one-host-thread", action="store_true", help="the learner half of a step enqueued by the thread that prepares the batches (default: by a second one)")
    ap.add_argument("--eager-blocks", action="store_true", help="the learner's module forward through PyTorch autograd under autocast instead of the fused Linear + BatchNorm + ReLU blocks (include/hz_train.h)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import bench
        sys.exit(bench.spawn_ranks(args.gpus, script=__file__))

    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    from hanabizero_amd.config import make_config
    from hanabizero_amd.device_replay import DeviceReplay
    from hanabizero_amd.dist import broadcast_weights, gather_packed
    from hanabizero_amd.learner import LearnerPipeline
    from hanabizero_amd.selfplay import SelfPlayActor

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            assert args.share_device or torch.cuda.device_count() >= world
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_dev = device if (world == 1 or args.backend == "nccl") else torch.device("cpu")
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    cfg = make_config(args.game, simulations=args.simulations, stack=4, p_mcts_num=args.envs, batch_size=args.batch_size)
    if args.checkpoint_interval:
        cfg.checkpoint_interval = args.checkpoint_interval
    if args.target_interval:
        cfg.target_model_interval = args.target_interval
    learns = rank == 0
    acts = world == 1 or rank > 0 or args.learner_acts
    n_actors = world if (world == 1 or args.learner_acts) else world - 1
    actor_index = rank if (world == 1 or args.learner_acts) else rank - 1

    engine = actor = None
    if acts:
        engine = bench.build_engine(cfg, dtype, device)
        actor_stream = None if args.actor_stream == "null" else torch.cuda.Stream(device=device, priority=0 if args.actor_stream == "normal" else 1)
        actor = SelfPlayActor(cfg, engine, args.envs, rank=actor_index, seed=0, device=device, stream=actor_stream)
        net_dev = cfg.get_uniform_network().to(device)  # the module the broadcast weights land in (in place, on the device)
        net_dev.eval()
    pipe = replay = None
    handover = {"pending": False, "count": 0, "ms": []}
    if learns:
        replay = DeviceReplay(cfg, args.replay_capacity, device=device)
        target = bench.build_engine(cfg, dtype, device)
        learner = cfg.get_uniform_network().to(device)
        learner.load_state_dict((engine or target)._net.state_dict())
        if not args.eager_blocks:
            from hanabizero_amd.fused_train import FusedTrainNet
            learner = FusedTrainNet(learner, unroll_steps=cfg.num_unroll_steps, parallel_heads=args.parallel_heads)

        def on_checkpoint(step, done_event):
            handover["pending"], handover["event"] = True, done_event
        pipe = LearnerPipeline(cfg, replay, learner, target, batch_size=cfg.batch_size, reanalyze_share=args.reanalyze_share,
                               on_checkpoint=on_checkpoint, host_thread=not args.one_host_thread, research_streams=args.research_streams, slots=args.slots or None)
    A = cfg.action_space_size
    W = (cfg.obs_dim + 31) // 32

    def barrier():
        if world > 1:
            dist.barrier()

    t_host = dict(selfplay=0.0, drain_gather=0.0, ingest=0.0, learner_enqueue=0.0, weights=0.0)
    totals = dict(moves=0, games=0)

    def hand_weights_over():
        """Every acting rank ends up with the learner's current weights in its engine, in place."""
        t0 = time.perf_counter()
        if world == 1:
            ws = actor._work_stream()
            with torch.cuda.stream(ws):
                ws.wait_event(handover["event"])
                engine.load(pipe.net)
                taken = torch.cuda.Event()
                taken.record(ws)
            pipe.flush()
            pipe.learn.wait_event(taken)  # (the next update must not overwrite what is being folded)
        else:
            state = None
            if learns:
                pipe.flush()
                pipe.learn.synchronize()
                state = {k: v.detach() for k, v in pipe.net.state_dict().items()}
            else:
                state = {k: v.detach() for k, v in net_dev.state_dict().items()}
            if coll_dev.type == "cpu":
                state = {k: v.cpu() for k, v in state.items()}
            got = broadcast_weights(state, src=0, device=coll_dev)
            if acts:
                with torch.cuda.stream(actor._work_stream()):
                    net_dev.load_state_dict({k: v.to(device) for k, v in got.items()})
                    engine.load(net_dev)
        handover["pending"] = False
        handover["count"] += 1
        handover["ms"].append(1e3 * (time.perf_counter() - t0))

    def one_round(train):
        t0 = time.perf_counter()
        packed = None
        if acts:
            for _ in range(args.flush_every):
                actor.step()
            packed = actor.drain_packed()   # (blocks for this rank's lock-steps; the learner's streams keep running)
        t1 = time.perf_counter()
        if learns and one_round.ingested is not None:  # (the receive buffers are reused: the last round's ingest has to have read them)
            torch.cuda.current_stream(device).wait_event(one_round.ingested)
        got = gather_packed(packed, A, W, dst=0, to_host=False)
        t2 = time.perf_counter()
        round_moves = torch.tensor([args.flush_every * args.envs if acts else 0], dtype=torch.int64, device=coll_dev)
        if world > 1:
            dist.all_reduce(round_moves)
        moves = int(round_moves)
        steps = 0
        if learns:
            with torch.cuda.stream(pipe.prep):
                pipe.prep.wait_stream(torch.cuda.current_stream(device))
                for buf, n, mv in got:
                    replay.ingest_packed(buf, n, mv)
                    totals["games"] += n
                one_round.ingested = torch.cuda.Event()
                one_round.ingested.record(pipe.prep)
            t3 = time.perf_counter()
            if train:
                one_round.owed += moves * args.ratio
                while one_round.owed >= 1.0 and (not args.max_steps_per_round or steps < args.max_steps_per_round):
                    pipe.step()
                    steps += 1
                    one_round.owed -= 1.0
                    if handover["pending"] and world == 1:
                        hand_weights_over()
                if pipe.steps and pipe.steps % 200 < steps:
                    replay.remove_to_fit()  # train.py:367-368
            t4 = time.perf_counter()
            t_host["ingest"] += t3 - t2
            t_host["learner_enqueue"] += t4 - t3
        if world > 1:  # the round's closing exchange: (trained steps, weights follow)
            flag = torch.tensor([pipe.steps if learns else 0, int(handover["pending"]) if learns else 0], dtype=torch.int64, device=coll_dev)
            dist.broadcast(flag, src=0)
            if int(flag[1]):
                hand_weights_over()
        t_host["selfplay"] += t1 - t0
        t_host["drain_gather"] += t2 - t1
        totals["moves"] += moves
        return steps
    one_round.owed = 0.0
    one_round.ingested = None

    for _ in range(args.warm_rounds):
        one_round(train=False)
    if learns and args.ratio > 0:
        assert replay.get_total_len() > cfg.batch_size, "the warm rounds finished too few games (%d positions)" % replay.get_total_len()
        for _ in range(3):   # (first shapes of the reanalyze search, hipBLASLt workspaces)
            pipe.step()
        pipe.flush()
        pipe.learn.synchronize()
    torch.cuda.synchronize(device)
    barrier()
    for k in t_host:
        t_host[k] = 0.0
    totals.update(moves=0, games=0)
    steps0 = pipe.steps if learns else 0
    t_start = time.perf_counter()
    for r in range(args.rounds):
        one_round(train=args.ratio > 0)
    if learns:
        pipe.flush()
        pipe.learn.synchronize()
        pipe.prep.synchronize()
    torch.cuda.synchronize(device)
    barrier()
    wall = time.perf_counter() - t_start
    wall_t = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t)
    # one weight hand-over timed on its own
    if learns:
        handover["event"] = pipe.slots[(pipe.steps - 1) % len(pipe.slots)].done if pipe.steps else torch.cuda.Event()
        if not pipe.steps:
            handover["event"].record(pipe.learn)
    handover_each = []
    for _ in range(3):   # (the first one of a process also builds the fragment-gather indices on the device: reported apart)
        torch.cuda.synchronize(device)
        barrier()
        t0 = time.perf_counter()
        hand_weights_over()
        torch.cuda.synchronize(device)
        handover_each.append(1e3 * (time.perf_counter() - t0))
    handover_ms = min(handover_each[1:])
    if acts:
        for _ in range(3):
            actor.step()
        torch.cuda.synchronize(device)
        assert int(actor.illegal_steps) == 0
    if learns:
        steps_done = pipe.steps - steps0
        losses = pipe.losses() if pipe.steps else None
        out = {"workload": "%s, %d envs x %d acting rank(s), %d sims/move, %s nets: self-play + reanalyze (%.0f %% of each batch) + learner batch %d; "
                           "replay, batch maker and learner on the device of rank 0" % (args.game, args.envs, n_actors, args.simulations, args.dtype,
                                                                                       100 * args.reanalyze_share, cfg.batch_size),
               "n_gpus": world, "backend": args.backend if world > 1 else None, "share_device": bool(args.share_device),
               "roles": {"learner_rank": 0, "acting_ranks": n_actors, "learner_acts": bool(world == 1 or args.learner_acts)},
               "rounds": args.rounds, "lock_steps_per_round": args.flush_every,
               "selfplay_moves_per_s": totals["moves"] / wall, "learner_steps_per_s": steps_done / wall, "learner_steps": steps_done,
               "replay_ratio_target": args.ratio, "replay_ratio_achieved": steps_done / max(1, totals["moves"]),
               "reference": {"replay_ratio": 0.008, "learner_steps_per_s": 1000 / 160.0, "selfplay_moves_per_s_derived": 1000 / 160.0 / 0.008,
                             "hardware": "4 x RTX 3090 + 96 CPU cores, Hanabi-Small", "source": "/root/reference/README.md:51"},
               "wall_s": wall, "games_ingested": totals["games"], "replay_positions": replay.get_total_len(), "replay_hbm_bytes": replay.hbm_bytes,
               "host_ms_per_round": {k: 1e3 * v / args.rounds for k, v in t_host.items()},
               "host_ms_per_learner_step_enqueue": (1e3 * t_host["learner_enqueue"] / steps_done) if steps_done else None,
               "weight_handover_ms": handover_ms, "weight_handover_first_ms": handover_each[0], "weight_handovers_in_run": handover["count"] - 3,
               "host_wait_for_the_gpu_ms_per_learner_step": (1e3 * pipe.host_wait_s / max(1, pipe.steps)),
               "checkpoint_interval": cfg.checkpoint_interval, "target_model_interval": cfg.target_model_interval,
               "loss_last": losses[1] if losses else None, "learner_blocks": "autograd (autocast)" if args.eager_blocks else "fused (include/hz_train.h)",
               "prepare_stream_candidates_ms": pipe.prep_interference_ms, "stream_probe_ms": pipe.stream_probe_ms,
               "host_threads_on_the_learner_rank": 1 if args.one_host_thread else 2}
        print(json.dumps(out), flush=True)
        pipe.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
