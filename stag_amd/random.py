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
# xor-ed into the seed of streams that are not edge noise (GAT's attention-dropout mask): same generator, same
# offset counter, a different key — the two can never draw from one (seed, offset) pair
ATTN_DROP_DOMAIN = 0xA77D209D0F5EED55


class NoiseGenerator:
    def __init__(self, seed=None):
        self._lock = threading.Lock()
        self.manual_seed(torch.initial_seed() if seed is None else seed)

    def manual_seed(self, seed):
        with self._lock:
            self.seed = int(seed) & _MASK64
            self.offset = 0
            if getattr(self, "_epoch", None) is not None:
                self._epoch.zero_()
        return self

    # ---- hipGraph support ------------------------------------------------------------
    # A captured graph freezes the host-side offsets of its kernels.  With a device epoch the
    # kernels draw at offset + *epoch, and `advance_epoch(k)` — a device-side add that is captured
    # like any other node — moves every replay to fresh noise:
    #
    #     gen.enable_device_epoch(device)
    #     with torch.cuda.graph(g):
    #         mark = gen.offset
    #         loss = model.loss(graph, x, y); loss.backward(); opt.step()
    #         gen.advance_epoch(gen.offset - mark)
    #     for _ in range(steps): g.replay()
    def enable_device_epoch(self, device):
        """Allocate the device counter; every EdgeNoise drawn from this generator carries it."""
        self._epoch = torch.zeros(1, dtype=torch.int64, device=device)
        return self._epoch

    @property
    def device_epoch(self):
        return getattr(self, "_epoch", None)

    def advance_epoch(self, k=1):
        """epoch += k on the device (stream-ordered; capturable)."""
        if getattr(self, "_epoch", None) is None:
            raise RuntimeError("enable_device_epoch(device) first")
        self._epoch.add_(int(k))

    def next_offset(self):
        with self._lock:
            o = self.offset
            self.offset = (self.offset + 1) & _MASK64
        return o

    def get_state(self):
        """Host state; add `int(device_epoch)` to the offset yourself when a graph is in use."""
        return {"seed": self.seed, "offset": self.offset}

    def set_state(self, state):
        with self._lock:
            self.seed, self.offset = int(state["seed"]) & _MASK64, int(state["offset"]) & _MASK64


default_generator = NoiseGenerator(seed=0x5747A6)


def manual_seed(seed):
    """Seed the edge-noise stream (all ranks of a partitioned run must use one seed)."""
    return default_generator.manual_seed(seed)
