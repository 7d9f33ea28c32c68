"""tests/hip_adapters.py -- give the HIP library (through hanabizero_amd's ctypes layer, i.e. through the C ABI) the
same method set as oracle.cport.OracleTree / OracleEnv so one scenario runner drives all implementations."""
import numpy as np
import torch


class HipTree:
    def __init__(self, N, A, S, seed=0, value_delta_max=0.006, tree_id_base=0):
        from hanabizero_amd import cytree
        self.N, self.A, self.S = N, A, S
        self.roots = cytree.Roots(N, A, S, tie_seed=seed, tree_id_base=tree_id_base)
        self.delta = value_delta_max

    def prepare(self, frac, noises, rewards, logits, legal):
        self.roots.prepare(frac, np.asarray(noises, np.float32), np.asarray(rewards, np.float32),
                           np.asarray(logits, np.float32), np.asarray(legal, np.uint8))

    def prepare_no_noise(self, rewards, logits, legal):
        self.roots.prepare_no_noise(np.asarray(rewards, np.float32), np.asarray(logits, np.float32),
                                    np.asarray(legal, np.uint8))

    def traverse(self, sim, pb_c_base, pb_c_init, discount):
        self.roots.set_params(pb_c_base, pb_c_init, discount, self.delta)
        assert self.roots._sim == sim
        ix, iy, la = self.roots.traverse_tensors()
        return ix.cpu().numpy(), iy.cpu().numpy(), la.cpu().numpy()

    def path_len(self):
        return self.roots.path_len_tensor().cpu().numpy()

    def backprop(self, hidden_state_index_x, discount, rewards, values, logits):
        self.roots.backprop_tensors(hidden_state_index_x, np.asarray(rewards, np.float32),
                                    np.asarray(values, np.float32), np.asarray(logits, np.float32))

    def distributions(self):
        return self.roots.distributions_tensor().cpu().numpy()

    def values(self):
        return self.roots.values_tensor().cpu().numpy()

    def trajectories(self, max_len=None):
        return self.roots.trajectories_tensor(max_len or self.S).cpu().numpy()

    def minmax(self):
        mn, mx = self.roots.minmax_tensors()
        return mn.cpu().numpy(), mx.cpu().numpy()

    def root_priors(self):
        return self.roots.root_priors_tensor().cpu().numpy()


def sync():
    torch.cuda.synchronize()


class HipEnv:
    """OracleEnv-shaped adapter over hanabizero_amd.hanabi_env.HanabiVecEnv (global observation)."""

    def __init__(self, name, seeds):
        from hanabizero_amd.hanabi_env import HanabiVecEnv
        self.e = HanabiVecEnv(name, seeds, mdp="global")
        self.N, self.num_moves, self.obs_len, self.own_len, self.players = (
            self.e.N, self.e.num_moves, self.e.obs_len, self.e.own_len, self.e.players)
        self.D = self.e.obs_dim

    def reset(self, mask=None):
        self.e.reset(None if mask is None else np.asarray(mask, np.uint8))

    def step(self, actions, mask=None):
        r, d, s, st = self.e.step(np.asarray(actions, np.int32), None if mask is None else np.asarray(mask, np.uint8))
        st = st.cpu().numpy()
        m = np.ones(self.N, bool) if mask is None else np.asarray(mask, bool)
        if (st[m] != 0).any():
            raise ValueError("illegal move in env %d" % int(np.nonzero((st != 0) & m)[0][0]))
        return r.cpu().numpy(), d.cpu().numpy(), s.cpu().numpy()

    def observe(self):
        obs, legal = self.e.observe()
        return obs.cpu().numpy(), legal.cpu().numpy()

    def probe(self):
        return self.e.probe().cpu().numpy()
