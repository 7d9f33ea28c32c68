"""tests/netgold.py -- deterministic weight recipe shared by tools/gen_golden.py and tests/test_model.py.

The nets fixtures hold only inputs and the reference's outputs; the (large) state_dict is regenerated from this
recipe on both sides, keyed by parameter name, so it does not have to be committed.
"""
import zlib

import numpy as np


def make_weight(key, shape):
    rng = np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, np.int64)
    if key.endswith("running_var"):
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if key.endswith("running_mean"):
        return rng.normal(0, 0.2, shape).astype(np.float32)
    if key.endswith("bias"):
        return rng.normal(0, 0.1, shape).astype(np.float32)
    if len(shape) == 2:  # Linear.weight [out, in]
        return rng.normal(0, 1.0 / np.sqrt(shape[1]), shape).astype(np.float32)
    return rng.uniform(0.8, 1.2, shape).astype(np.float32)  # BatchNorm1d.weight


def fill_state_dict(state_dict):
    """Returns {key: numpy array} for every entry of a torch state_dict (shapes taken from it)."""
    return {k: make_weight(k, tuple(v.shape)) for k, v in state_dict.items()}
