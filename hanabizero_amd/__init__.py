"""hanabizero_amd -- MI355X-native batched self-play + MCTS engine behind HanabiZero's own interfaces.

Only what the hot path needs: the HIP kernels + C ABI (csrc/, include/), and the host-side mirrors of the
reference interfaces that call them (cytree, mcts, hanabi_env, game, selfplay, model, config).
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is missing: no CPU fallback)

__all__ = ["_lib"]
