"""Seed / offset state of the counter-based edge-noise stream.

The reference draws from torch's global generator (stag/layers.py:123-127); a
Philox counter stream needs (seed, offset) instead.  Every fused draw of an
[E, Dn] noise field consumes one offset, the way a torch CUDA generator advances
its Philox offset per kernel, so successive layers / Monte-Carlo samples
(stag/models.py:45-55, 67-68) get independent noise and a fixed seed replays a run.
"""
import threading

import torch

_MASK64 = (1 << 64) - 1


class NoiseGenerator:
    def __init__(self, seed=None):
        self._lock = threading.Lock()
        self.manual_seed(torch.initial_seed() if seed is None else seed)

    def manual_seed(self, seed):
        with self._lock:
            self.seed = int(seed) & _MASK64
            self.offset = 0
        return self

    def next_offset(self):
        with self._lock:
            o = self.offset
            self.offset = (self.offset + 1) & _MASK64
        return o

    def get_state(self):
        return {"seed": self.seed, "offset": self.offset}

    def set_state(self, state):
        with self._lock:
            self.seed, self.offset = int(state["seed"]) & _MASK64, int(state["offset"]) & _MASK64


default_generator = NoiseGenerator(seed=0x5747A6)


def manual_seed(seed):
    """Seed the edge-noise stream (all ranks of a partitioned run must use one seed)."""
    return default_generator.manual_seed(seed)
