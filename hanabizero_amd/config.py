"""hanabizero_amd.config -- ``config/hanabi_control`` for the MI355X build.

The attributes the self-play / search / evaluation path reads, with the reference's names and values
(/root/reference/core/config.py:20-131, config/hanabi_control/__init__.py:10-245), plus ``new_game`` /
``get_uniform_network`` / ``set_game`` / transforms.  Learner-only hyper-parameters are carried so that
core/train.py finds them, but nothing here schedules a learner.  ``HanabiControlConfigFull`` defaults the
``rmsprop`` flag the reference's main.py forgets to define (config/hanabi_control/__init__.py:185).
"""
import types

import numpy as np
import torch

from .model import MuZeroNet, MuZeroNetFull, inverse_scalar_transform


class DiscreteSupport:  # core/config.py:9-16
    def __init__(self, min, max, delta=1.):
        assert min < max
        self.min, self.max, self.delta = min, max, delta
        self.range = np.arange(min, max + 1, delta)
        self.size = len(self.range)


_DEFAULT_ARGS = dict(simulations=50, batch_size=256, td_steps=5, actors=1, lr=0.1, decay_rate=1.0, decay_step=200000,
                     stack=4, const=0.0, val_coeff=0.25, debug_batch=False, debug_interval=100, rmsprop=0,
                     p_mcts_num=4096, seed=0, mdp_type="global", amp_type="torch_amp", env="Hanabi-Full",
                     use_priority=True, use_max_priority=True)


def make_args(**kw):
    d = dict(_DEFAULT_ARGS)
    d.update(kw)
    return types.SimpleNamespace(**d)


class _HanabiConfig:
    def __init__(self, args, max_moves, support, training_steps, last_steps, checkpoint_interval, test_episodes,
                 lr_warm_up=0.001):
        # core/config.py:95-131 and the per-game constructors
        self.num_simulations = args.simulations
        self.batch_size, self.td_steps, self.num_actors = args.batch_size, args.td_steps, args.actors
        self.num_unroll_steps = 5
        self.max_moves = self.test_max_moves = max_moves
        self.history_length = 12001
        self.discount = 0.999
        self.value_delta_max = 0.006
        self.root_dirichlet_alpha = 0.3
        self.root_exploration_fraction = 0.25
        self.pb_c_base, self.pb_c_init = 19652, 1.25
        self.stacked_observations = args.stack
        self.value_support = DiscreteSupport(-support, support, delta=1)
        self.reward_support = DiscreteSupport(-support, support, delta=1)
        self.training_steps, self.last_steps = training_steps, last_steps
        self.checkpoint_interval, self.test_episodes = checkpoint_interval, test_episodes
        self.target_model_interval = 200  # config/hanabi_control/__init__.py:20, 138 (both games)
        self.self_play_moves_ratio = 1
        self.clip_reward = self.image_based = self.cvt_string = self.state_norm = self.use_epsilon_greedy = False
        self.change_temperature = False
        self.init_zero = True
        self.prioritized_replay_eps = 1e-6
        # learner (core/config.py:98, 136, 153-161; config/hanabi_control/__init__.py:43, 161)
        self.max_grad_norm, self.weight_decay, self.momentum = 5, 1e-4, 0.9
        self.priority_reward_ratio = 0
        self.lr_warm_up = lr_warm_up
        self.lr_warm_step = int(training_steps * lr_warm_up)
        self.lr_init, self.lr_decay_rate, self.lr_decay_steps = args.lr, args.decay_rate, args.decay_step
        self.value_loss_coeff, self.reward_loss_coeff, self.policy_loss_coeff = args.val_coeff, 1, 1
        self.consistency_coeff = self.const = getattr(args, "const", 0.0)
        self.action_space_size = None
        self.obs_shape = None
        self.env_name = None
        self.mdp = "global"

    # core/config.py:262-300 (the part the hot path reads)
    def set_config(self, args):
        self.mdp = args.mdp_type
        self.set_game(args.env)
        self.seed = args.seed
        self.amp_type = args.amp_type
        self.use_priority = args.use_priority
        self.use_max_priority = args.use_max_priority if self.use_priority else False
        self.p_mcts_num = args.p_mcts_num
        return None

    def visit_softmax_temperature_fn(self, num_moves, trained_steps):
        return 1.0  # change_temperature is False in both configs (__init__.py:34,152)

    def set_game(self, env_name, **_):
        from .hanabi_env import GAMES
        self.env_name = env_name
        g = GAMES[env_name]
        hand = g["hand_size"] if g["hand_size"] > 0 else (5 if g["players"] < 4 else 4)
        c, r, p = g["colors"], g["ranks"], g["players"]
        per_color = sum(3 if k == 0 else (1 if k == r - 1 else 2) for k in range(r))
        obs = ((p - 1) * hand * c * r + p) + (per_color * c - p * hand + c * r + g["max_information_tokens"] +
                                              g["max_life_tokens"]) + per_color * c + \
              (p + 4 + p + c + r + hand + hand + c * r + 2) + p * hand * (c * r + c + r)
        own = hand * c * r
        self.obs_dim = (own if self.mdp == "global" else 0) + obs + p
        self.obs_shape = self.obs_dim * self.stacked_observations  # __init__.py:90-93
        self.action_space_size = 2 * hand + (p - 1) * c + (p - 1) * r

    def new_game(self, seed=None, **_):  # __init__.py:110-117, 228-234
        from .hanabi_env import HanabiControlWrapper, HanabiEnv
        return HanabiControlWrapper(HanabiEnv({"hanabi_name": self.env_name, "seed": seed}), discount=self.discount,
                                    mdp=self.mdp)

    def inverse_value_transform(self, logits):
        return inverse_scalar_transform(logits, self.value_support.min, self.value_support.max)

    def inverse_reward_transform(self, logits):
        return inverse_scalar_transform(logits, self.reward_support.min, self.reward_support.max)

    def scalar_reward_loss(self, prediction, target):
        return -(torch.log_softmax(prediction, dim=1) * target).sum(1)

    scalar_value_loss = scalar_reward_loss


class HanabiControlConfig(_HanabiConfig):  # Hanabi-Small (__init__.py:10-126)
    def __init__(self, args=None):
        args = args or make_args(env="Hanabi-Small")
        super().__init__(args, max_moves=60, support=25, training_steps=200000, last_steps=0,
                         checkpoint_interval=1000, test_episodes=40)

    def get_uniform_network(self):
        assert self.env_name == "Hanabi-Small"
        return MuZeroNet(self.obs_shape, self.action_space_size, self.reward_support.size, self.value_support.size,
                         self.inverse_value_transform, self.inverse_reward_transform, state_norm=self.state_norm)


class HanabiControlConfigFull(_HanabiConfig):  # Hanabi-Full (__init__.py:128-245)
    def __init__(self, args=None):
        args = args or make_args(env="Hanabi-Full")
        super().__init__(args, max_moves=160, support=100, training_steps=3000000, last_steps=100,
                         checkpoint_interval=2000, test_episodes=80, lr_warm_up=0.0001)

    def get_uniform_network(self):
        assert self.env_name in ("Hanabi-Full", "Hanabi-Full-5p")
        return MuZeroNetFull(self.obs_shape, self.action_space_size, self.reward_support.size, self.value_support.size,
                             self.inverse_value_transform, self.inverse_reward_transform, state_norm=self.state_norm)


def make_config(env_name, **kw):
    args = make_args(env=env_name, **kw)
    cfg = HanabiControlConfig(args) if env_name == "Hanabi-Small" else HanabiControlConfigFull(args)
    cfg.set_config(args)
    return cfg
