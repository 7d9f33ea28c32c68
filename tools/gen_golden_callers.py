#!/usr/bin/env python3
"""tools/gen_golden_callers.py -- fixtures for the PYTHON callers either side of the hot path, produced by the reference's
own Python (imported from /root/reference; authoring container only -- the reference never travels).

The reference's modules import ray / gym / cv2 / baselines / absl / the Cython cytree at module level; none of those is
installed here and none is needed by the functions recorded below, so inert stand-ins are registered in sys.modules for the
duration of this script (they are import plumbing of THIS generator, not part of any build; nothing of them is committed
as a reference stand-in).  What is recorded is data: inputs and the reference's outputs.

  select_action.npz        core/utils.py:280-295 -- visit counts / legal masks / numpy seeds -> (action, entropy), plus the
                           uniform np.random.choice consumed (the product's kernel takes that uniform as an input)
  game_history.npz         core/game.py:49-214 GameHistory (init / append / store_search_stats / step_obs / game_over / obs /
                           save_file) and DataWorker.put's turn-reward reshape (core/selfplay_worker.py:29-39)
  learner_step_<game>.npz  core/train.py:59-314 update_weights: one SGD step of the reference on a fixed batch from a fixed
                           state_dict (tests/netgold.py recipe), CPU fp32 (amp_type 'none'): losses, priorities, gradient
                           norms after clipping, digests of the updated parameters and BatchNorm statistics
  batch_targets_<game>.npz core/reanalyze_worker.py BatchWorker_CPU.make_batch (:148-204) + BatchWorker_GPU.
                           _prepare_reward_value (:249-304) + _prepare_policy_non_re (:374-399): inputs and targets of a
                           learner batch for fixed games / positions with the reference net as target model

Usage: python tools/gen_golden_callers.py [--only select_action|game_history|learner_step|batch_targets]
"""
import argparse
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------------------------------- import plumbing
def _install_stand_ins():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def remote(*args, **kwargs):  # @ray.remote and @ray.remote(num_gpus=...): the class / function itself
        if len(args) == 1 and not kwargs and (isinstance(args[0], type) or callable(args[0])):
            return args[0]
        return lambda obj: obj

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Anything()

        def __call__(self, *a, **k):
            return _Anything()

    ray = mod("ray", remote=remote, put=lambda x: x, get=lambda x: x, init=lambda *a, **k: None, wait=lambda x: (x, []))
    ray.util = mod("ray.util")
    ray.util.queue = mod("ray.util.queue", Queue=_Anything)
    ray.util.multiprocessing = mod("ray.util.multiprocessing", Pool=_Anything)
    mod("cv2", INTER_AREA=0)
    gym = mod("gym", Wrapper=object, ObservationWrapper=object, Env=object, make=_Anything())
    gym.spaces = mod("gym.spaces", Box=_Anything, Discrete=_Anything)
    mod("baselines")
    mod("baselines.common")
    mod("baselines.common.atari_wrappers", WarpFrame=_Anything, EpisodicLifeEnv=_Anything)
    mod("envs", HanabiEnv=_Anything)  # (the C++ env is pinned by tests/golden/env_*.npz through its C API)
    mod("absl", flags=_Anything())
    # the Cython tree is pinned through oracle/_ref (tests/golden/tree_*.npz); the Python recorded here never calls it
    mod("core.ctree.cytree", Node=_Anything, Roots=_Anything, MinMaxStatsList=_Anything, ResultsWrapper=_Anything)
    try:
        import tqdm  # noqa: F401
    except ImportError:
        mod("tqdm", tqdm=lambda x, *a, **k: x)
        mod("tqdm.auto", tqdm=lambda x, *a, **k: x)


def _reference():
    _install_stand_ins()
    sys.path.insert(0, REF)
    import core  # noqa: F401
    import core.ctree  # noqa: F401  (package; its cytree submodule is the stand-in above)
    import core.utils as utils
    import core.game as game
    import core.config as rconfig
    import core.selfplay_worker as spw
    import core.reanalyze_worker as rw
    import core.train as train
    import config.hanabi_control as hc
    return types.SimpleNamespace(utils=utils, game=game, config=rconfig, spw=spw, rw=rw, train=train, hc=hc)


def _ref_config(R, game_name, stack, batch_size):
    """HanabiControlConfig / HanabiControlConfigFull with the flags of train.sh (main.py:16-90 defaults), the fields
    set_config / set_game derive (core/config.py:262-300, config/hanabi_control/__init__.py:84-93) filled in by hand (set_game would
    build an env), everything on the CPU in fp32."""
    args = argparse.Namespace(simulations=50, batch_size=batch_size, td_steps=5, actors=1, lr=0.1, decay_rate=0.1, stack=stack,
                              const=0, val_coeff=0.25, rmsprop=0, num_unroll_steps=5,
                              decay_step=200000, debug_batch=False, debug_interval=1)
    full = game_name != "Hanabi-Small"
    cfg = (R.hc.HanabiControlConfigFull if full else R.hc.HanabiControlConfig)(args)
    D, A = {"Hanabi-Small": (193, 11), "Hanabi-Full": (785, 20)}[game_name]
    cfg.env_name, cfg.mdp = game_name, "global"
    cfg.obs_shape, cfg.action_space_size = D * stack, A
    cfg.device, cfg.amp_type = "cpu", "none"
    cfg.use_augmentation = False
    return cfg


def _fill(net):
    import torch
    from tests.netgold import fill_state_dict
    net.load_state_dict({k: torch.from_numpy(v) for k, v in fill_state_dict(net.state_dict()).items()})
    return net


# ------------------------------------------------------------------------------------------- select_action
def gen_select_action(R):
    rng = np.random.RandomState(11)
    counts, legal, seeds, det, acts, ents, us, temps = [], [], [], [], [], [], [], []
    for case in range(400):
        A = [11, 20, 48][case % 3]
        c = rng.randint(0, 12, A)
        c[rng.rand(A) < 0.3] = 0
        lg = (rng.rand(A) < 0.6).astype(np.int64)
        if case % 7 == 0:
            lg[:] = 1
        if (c * lg).sum() == 0:  # (the reference divides by the masked sum: keep it positive)
            j = rng.randint(A)
            c[j], lg[j] = 3, 1
        deterministic = case % 4 == 0
        T = [1.0, 1.0, 0.5, 0.25][case % 4] if case >= 200 else 1.0
        seed = 1000 + case
        np.random.seed(seed)
        a, e = R.utils.select_action(list(c), temperature=T, deterministic=deterministic, legal_actions=list(lg))
        u = np.random.RandomState(seed).random_sample()  # what np.random.choice drew (mtrand: cdf.searchsorted(u, 'right'))
        pad = lambda x: np.concatenate([x, np.zeros(48 - A, x.dtype)])
        counts.append(pad(c)), legal.append(pad(lg)), seeds.append(seed), det.append(deterministic)
        acts.append(int(a)), ents.append(float(e)), us.append(u), temps.append(T)
    np.savez_compressed(os.path.join(GOLD, "select_action.npz"), counts=np.array(counts), legal=np.array(legal),
                        num_actions=np.array([[11, 20, 48][i % 3] for i in range(400)]), seed=np.array(seeds),
                        deterministic=np.array(det), temperature=np.array(temps), uniform=np.array(us),
                        action=np.array(acts), entropy=np.array(ents))
    print("select_action.npz: 400 cases")


# ------------------------------------------------------------------------------------------- GameHistory + put
def _play(R, cfg, rng, T, terminal_in_obs=True):
    """A synthetic finished game through the reference GameHistory exactly as DataWorker.run_multi drives it
    (selfplay_worker.py:122-138, 300-330): init with `stack` copies of the first observation, then per move
    store_search_stats + append."""
    D, A, stack = cfg.obs_shape // cfg.stacked_observations, cfg.action_space_size, cfg.stacked_observations
    gh = R.game.GameHistory(None, max_length=cfg.max_moves, config=cfg)
    obs0 = (rng.rand(D) < 0.2).astype(np.int64)
    legal0 = (rng.rand(A) < 0.7).astype(np.float64)
    gh.init([obs0 for _ in range(stack)], legal0)
    raw = dict(obs=[obs0], legal=[legal0], action=[], reward=[], visits=[], value=[])
    for t in range(T):
        visits = rng.randint(0, 9, A)
        visits[rng.randint(A)] += 1
        value = float(rng.randn())
        a = int(rng.randint(A))
        o = (rng.rand(D) < 0.2).astype(np.int64)
        r = int(rng.randint(-2, 3))
        lg = (rng.rand(A) < 0.7).astype(np.float64)
        gh.store_search_stats(list(visits), value)
        gh.append(a, o, r, lg)
        raw["visits"].append(visits), raw["value"].append(value), raw["action"].append(a), raw["obs"].append(o)
        raw["reward"].append(r), raw["legal"].append(lg)
    return gh, {k: np.array(v) for k, v in raw.items()}


def gen_game_history(R):
    cfg = _ref_config(R, "Hanabi-Small", stack=3, batch_size=8)
    rng = np.random.RandomState(5)
    out = {}
    for g, T in enumerate([1, 7, 23]):
        gh, raw = _play(R, cfg, rng, T)
        step_obs_mid = np.array(gh.step_obs())  # the window the next root inference would see (game.py:168-173)
        gh.game_over()
        worker = R.spw.DataWorker.__new__(R.spw.DataWorker)
        worker.trajectory_pool = []
        worker.put((gh, None))  # turn-reward reshape, in place on gh.rewards
        saved = gh.save_file()
        for k, v in raw.items():
            out["g%d_in_%s" % (g, k)] = v
        for k in ("vis", "root", "a", "o", "r", "la"):
            out["g%d_out_%s" % (g, k)] = np.asarray(saved[k])
        out["g%d_len" % g] = len(gh)
        out["g%d_step_obs" % g] = step_obs_mid
        out["g%d_obs_1_2_pad" % g] = np.asarray(gh.obs(min(1, T), extra_len=2, padding=True))
        out["g%d_obs_last_5_pad" % g] = np.asarray(gh.obs(T, extra_len=5, padding=True))
        out["g%d_zero_obs" % g] = np.asarray(gh.zero_obs())
    out["stack"] = 3
    np.savez_compressed(os.path.join(GOLD, "game_history.npz"), **out)
    print("game_history.npz: 3 games")


# ------------------------------------------------------------------------------------------- learner step
def _digest(t):
    t = t.detach().double().reshape(-1)
    return np.array([float(t.sum()), float(t.abs().sum()), float((t * t).sum())] + [float(x) for x in t[:5]] +
                    [0.0] * max(0, 5 - t.numel()))


def gen_learner_step(R, game_name, stack, B):
    import torch
    cfg = _ref_config(R, game_name, stack, B)
    D, A, U = cfg.obs_shape // stack, cfg.action_space_size, cfg.num_unroll_steps
    torch.manual_seed(0)
    model = _fill(cfg.get_uniform_network())
    model.train()  # core/train.py:323
    opt = torch.optim.SGD(model.parameters(), lr=cfg.lr_init, momentum=cfg.momentum, weight_decay=cfg.weight_decay)  # :327
    rng = np.random.RandomState(3)
    obs = (rng.rand(B, stack + U, D) < 0.2).astype(np.float32)
    action = rng.randint(0, A, (B, U))
    mask = np.ones((B, U), np.float32)
    indices = np.arange(B)
    weights = rng.uniform(0.3, 1.0, B).astype(np.float32)
    make_time = np.zeros(B)
    target_reward = rng.randint(-2, 3, (B, U)).astype(np.float32)
    target_value = (rng.randn(B, U + 1) * 6).astype(np.float32)
    tp = rng.rand(B, U + 1, A).astype(np.float32)
    tp[rng.rand(B, U + 1) < 0.15] = 0  # past-the-end positions carry all-zero policy targets
    s = tp.sum(-1, keepdims=True)
    target_policy = np.where(s > 0, tp / np.maximum(s, 1e-9), 0).astype(np.float32)

    class Sink:  # replay_buffer.update_priorities.remote(indices, new_priority, make_time)  (train.py:252)
        def __init__(self):
            self.update_priorities = self
            self.got = None

        def remote(self, indices, prio, make_time):
            self.got = (np.asarray(indices), np.asarray(prio), np.asarray(make_time))
    sink = Sink()
    steps = 2  # the second step exercises momentum and the BatchNorm running statistics of the first
    res = {}
    for it in range(steps):
        batch = ([obs, action, mask, indices, weights, make_time], [target_reward, target_value, target_policy])
        loss_data, _, _, _ = R.train.update_weights(model, batch, opt, sink, cfg, None, False)
        res["loss_data_%d" % it] = np.array([float(x) for x in loss_data], np.float64)
        res["priority_%d" % it] = sink.got[1].astype(np.float64)
        names = [n for n, _ in model.named_parameters()]
        res["grad_norm_%d" % it] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
        res["param_digest_%d" % it] = np.stack([_digest(p) for _, p in model.named_parameters()])
        bufs = [(n, b) for n, b in model.named_buffers() if b.dtype.is_floating_point]
        res["buffer_digest_%d" % it] = np.stack([_digest(b) for _, b in bufs])
    res.update(obs=obs.astype(np.uint8), action=action, mask=mask, indices=indices, weights=weights, target_reward=target_reward,
               target_value=target_value, target_policy=target_policy, stack=stack, D=D, A=A, U=U,
               param_names=np.array(names), buffer_names=np.array([n for n, _ in bufs]),
               lr=cfg.lr_init, momentum=cfg.momentum, weight_decay=cfg.weight_decay, max_grad_norm=cfg.max_grad_norm,
               value_loss_coeff=cfg.value_loss_coeff, priority_reward_ratio=cfg.priority_reward_ratio,
               prioritized_replay_eps=cfg.prioritized_replay_eps, td_steps=cfg.td_steps, discount=cfg.discount)
    np.savez_compressed(os.path.join(GOLD, "learner_step_%s.npz" % game_name), **res)
    print("learner_step_%s.npz: %d steps, loss %s" % (game_name, steps, res["loss_data_0"][:3]))


# ------------------------------------------------------------------------------------------- batch inputs + targets
def gen_batch_targets(R, game_name, stack):
    import torch
    B = 12
    cfg = _ref_config(R, game_name, stack, B)
    cfg.target_infer_size = 5  # (slices of the target-model inference: reanalyze_worker.py:258-272)
    D, A, U, td = cfg.obs_shape // stack, cfg.action_space_size, cfg.num_unroll_steps, cfg.td_steps
    rng = np.random.RandomState(9)
    games, raws = [], []
    for T in (3, 9, 14, 30):
        gh, raw = _play(R, cfg, rng, T)
        gh.game_over()
        games.append(gh), raws.append(raw)
    pick = rng.randint(0, len(games), B)
    pick[:4] = [0, 1, 2, 3]
    game_lst = [games[i] for i in pick]
    pos_lst = [int(rng.randint(0, len(g))) for g in game_lst]
    pos_lst[0], pos_lst[1] = len(game_lst[0]) - 1, 0  # the last position of a short game; the first of another
    weights_lst = rng.uniform(0.2, 1.0, B)

    class Holder:  # mcts_storage.push(context) (reanalyze_worker.py:202)
        def push(self, ctx):
            self.ctx = ctx

    class Rb:  # replay_buffer.get_total_len.remote()
        def __init__(self):
            self.get_total_len = self

        def remote(self):
            return 12345
    cpu = R.rw.BatchWorker_CPU.__new__(R.rw.BatchWorker_CPU)
    cpu.config, cpu.replay_buffer, cpu.mcts_storage = cfg, Rb(), Holder()
    np.random.seed(77)  # the random actions that pad a window past the end of its game (reanalyze_worker.py:160)
    cpu.make_batch((game_lst, pos_lst, list(range(B)), weights_lst, [0.0] * B), 0.0, weights=None)
    reward_value_context, policy_re_context, policy_non_re_context, inputs_batch, _ = cpu.mcts_storage.ctx
    assert policy_re_context is None
    gpu = R.rw.BatchWorker_GPU.__new__(R.rw.BatchWorker_GPU)
    gpu.config = cfg
    torch.manual_seed(0)
    gpu.model = _fill(cfg.get_uniform_network())
    gpu.model.eval()
    batch_values, batch_rewards = gpu._prepare_reward_value(reward_value_context)
    batch_policies = gpu._prepare_policy_non_re(policy_non_re_context)
    # the context of the reanalyzed policy targets for the first 5 positions (reanalyze_worker.py:101-144): what the search
    # of _prepare_policy_re is prepared from (that search itself is pinned through the tree goldens + tests/test_callers.py)
    re_num = 5
    re_ctx = cpu._prepare_policy_re_context(list(range(re_num)), game_lst[:re_num], pos_lst[:re_num])
    out = dict(stack=stack, D=D, A=A, U=U, td_steps=td, discount=cfg.discount, pick=pick, positions=np.array(pos_lst),
               weights=weights_lst, in_obs=np.asarray(inputs_batch[0]).astype(np.uint8), in_action=np.asarray(inputs_batch[1]),
               in_mask=np.asarray(inputs_batch[2]), target_value=np.asarray(batch_values, np.float64),
               target_reward=np.asarray(batch_rewards, np.float64), target_policy=np.asarray(batch_policies, np.float64),
               pad_action_seed=77, re_num=re_num, re_obs=np.asarray(re_ctx[0]).astype(np.uint8), re_mask=np.asarray(re_ctx[1]),
               re_state_index=np.asarray(re_ctx[2]), re_indices=np.asarray(re_ctx[3]), re_traj_lens=np.asarray(re_ctx[5]),
               re_legal=np.asarray(re_ctx[6], np.float64))
    for i, raw in enumerate(raws):
        for k, v in raw.items():
            out["game%d_%s" % (i, k)] = v
    np.savez_compressed(os.path.join(GOLD, "batch_targets_%s.npz" % game_name), **out)
    print("batch_targets_%s.npz: B=%d, value range [%.2f, %.2f]" % (game_name, B, batch_values.min(), batch_values.max()))


def main():
    assert os.path.exists(os.path.join(REF, "core", "train.py")), "needs /root/reference (authoring container)"
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    R = _reference()
    if only in (None, "select_action"):
        gen_select_action(R)
    if only in (None, "game_history"):
        gen_game_history(R)
    if only in (None, "learner_step"):
        gen_learner_step(R, "Hanabi-Small", stack=2, B=8)
        gen_learner_step(R, "Hanabi-Full", stack=1, B=4)
    if only in (None, "batch_targets"):
        gen_batch_targets(R, "Hanabi-Small", stack=2)


if __name__ == "__main__":
    main()
