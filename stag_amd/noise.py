"""EdgeNoise: the [E, Dn] multiplicative edge weight of a StagLayer, kept lazy.

The reference materialises `q_a.expand([E, Dn]).sample()` and hands the tensor to
`base_layer.forward(edge_weight=...)` (stag/layers.py:96-113).  Here the handle
that travels is this descriptor: the fused kernel redraws the weights from
(seed, offset, CSR position, channel) on the fly, and `materialize()` produces the
reference's tensor only when somebody asks for it (`_edge_weight_sample`,
the sample-based KL fallback of stag/layers.py:142).
"""
import torch

from . import _lib

_KIND_OF = {
    torch.distributions.Normal: _lib.NOISE_NORMAL,
    torch.distributions.Uniform: _lib.NOISE_UNIFORM,
    torch.distributions.Bernoulli: _lib.NOISE_BERNOULLI,
}


def fusable(dist):
    """True if `dist` (a torch distribution) has an in-kernel sampler."""
    return type(dist) in _KIND_OF


def _param_mode(p, n_edges, dn):
    """Classify how a parameter tensor broadcasts against [E, Dn] (`.expand([E, Dn])`)."""
    if p.dim() == 2 and p.shape == (n_edges, 1) and n_edges == 1:
        return _lib.PARAM_PER_EDGE1          # the [E, 1] head of a one-edge graph is still one value per edge
    if p.dim() == 0 or p.numel() == 1:
        return _lib.PARAM_SCALAR
    if p.dim() == 1 and p.shape[0] == dn:
        return _lib.PARAM_PER_CHANNEL
    if p.dim() == 2 and p.shape[0] == n_edges and p.shape[1] == 1:
        return _lib.PARAM_PER_EDGE1
    if p.dim() == 2 and p.shape == (n_edges, dn):
        return _lib.PARAM_PER_EDGE
    if p.dim() == 2 and p.shape[0] == 1 and p.shape[1] == dn:
        return _lib.PARAM_PER_CHANNEL
    raise ValueError(f"parameter of shape {tuple(p.shape)} does not broadcast to [{n_edges}, {dn}]")


_ROW_CACHE = {}      # id(parameter tensor) -> (weakref to it, its version, {(shape, device): expanded row})


def _expanded(p, shape, dev, cache=False, src=None, tag=False):
    """p broadcast to `shape` as a contiguous device tensor.  For a per-channel row made from a module's buffer
    (a device scalar that must not be read back) the result is kept per source tensor while that tensor is alive
    and at the same version: two small launches less per layer call on launch-bound graphs."""
    if not (cache and torch.is_tensor(src) and not src.requires_grad):
        return p.detach().to(dev).expand(shape).contiguous()
    import weakref
    ent = _ROW_CACHE.get(id(src))
    if ent is None or ent[0]() is not src or ent[1] != src._version:
        if len(_ROW_CACHE) > 256:
            _ROW_CACHE.clear()
        ent = (weakref.ref(src), src._version, {})
        _ROW_CACHE[id(src)] = ent
    key = (tuple(shape), str(dev), bool(tag))        # tag: p = exp(src), not src
    row = ent[2].get(key)
    if row is None:
        row = p.detach().to(dev).expand(shape).contiguous()
        ent[2][key] = row
    return row


class EdgeNoise:
    def __init__(self, graph, dn, kind, p0, p1=None, relu=False, in_norm=False, seed=0, offset=0,
                 pos_base=0, differentiable=False, chunk_base=0, epoch=None, p1_log=False):
        # the owner of the cached structure, not a local_var() copy (whose frames hold step tensors)
        self.graph = graph._cache_owner() if hasattr(graph, "_cache_owner") else graph
        self.dn, self.kind = int(dn), int(kind)
        # vi=True: keep the live parameter tensors so ops.aggregate can return their gradients
        # (reparameterised draw; the backward regenerates the noise with spec.deriv = 1 | 2)
        self.grad_params = None
        if differentiable and kind in (_lib.NOISE_NORMAL, _lib.NOISE_UNIFORM):
            self.grad_params = (p0, p1)
        self.deriv = 0
        # p1 is log(scale) (Normal): the kernels exponentiate it where they load it and gradients come back
        # w.r.t. the log — what an AmortizedDistribution's [E, Dn] `log_scale` head wants (no exp pass, no saved tensor)
        self.p1_log = bool(p1_log) and kind == _lib.NOISE_NORMAL
        self.relu, self.in_norm = bool(relu), bool(in_norm)
        self.seed, self.offset, self.pos_base = int(seed), int(offset), int(pos_base)
        # channel shards (partition.ChannelShard): this tensor's channel 0 is global channel 4*chunk_base
        self.chunk_base = int(chunk_base)
        # device counter added to `offset` when the kernel runs (random.NoiseGenerator.device_epoch):
        # what makes a captured hipGraph draw fresh noise on every replay
        self.epoch = epoch
        # Monte-Carlo batching (StagLayer.forward_mc): n_samples draws at offset + s * offset_stride,
        # ops.aggregate then returns [n_samples, N, D] from shared gathers
        self.n_samples, self.offset_stride = 1, 1
        E = graph.number_of_edges()
        dev = graph.device
        self.param_mode = _lib.PARAM_SCALAR
        self.p0 = self.p1 = None
        self.p0_scalar = self.p1_scalar = 0.0
        if kind >= _lib.NOISE_NORMAL and all(isinstance(p, (int, float)) for p in ((p0,) if p1 is None else (p0, p1))):
            self.p0_scalar = float(p0)                  # plain numbers: no tensor round trip
            self.p1_scalar = float(p1) if p1 is not None else 0.0
        elif kind >= _lib.NOISE_NORMAL:
            ps = [torch.as_tensor(p, dtype=torch.float32) for p in ((p0,) if p1 is None else (p0, p1))]
            mode = max(_param_mode(p, E, dn) for p in ps)
            if mode == _lib.PARAM_SCALAR and any(p.is_cuda for p in ps):
                # a scalar that lives on the device (a module's buffer / parameter): reading it
                # back would stall the stream on every layer call and cannot be captured in a
                # hipGraph, so it travels as a per-channel row instead
                mode = _lib.PARAM_PER_CHANNEL
            self.param_mode = mode
            exp_applied = False
            if self.p1_log and mode == _lib.PARAM_PER_CHANNEL:      # the library takes log-scales per edge or scalar
                ps[1] = ps[1].exp()
                exp_applied = True
                self.p1_log = False
                if self.grad_params is not None:
                    raise ValueError("a per-channel log-scale with gradients: exponentiate it yourself (p1_log=False)")
            if mode == _lib.PARAM_SCALAR:
                self.p0_scalar = float(ps[0].detach().reshape(()))
                if p1 is not None:
                    self.p1_scalar = float(ps[1].detach().reshape(()))
            else:
                shape = {_lib.PARAM_PER_CHANNEL: (dn,), _lib.PARAM_PER_EDGE1: (E, 1),
                         _lib.PARAM_PER_EDGE: (E, dn)}[mode]
                srcs = (p0,) if p1 is None else (p0, p1)
                ex = [_expanded(p, shape, dev, cache=(mode == _lib.PARAM_PER_CHANNEL), src=srcs[i],
                                tag=(i == 1 and exp_applied)) for i, p in enumerate(ps)]
                self.p0 = ex[0]
                self.p1 = ex[1] if p1 is not None else None

    @classmethod
    def from_distribution(cls, graph, dn, dist, **kw):
        """dist: torch Normal / Uniform / Bernoulli with scalar, [Dn], [E,1] or [E,Dn] params."""
        kind = _KIND_OF[type(dist)]
        if kind == _lib.NOISE_NORMAL:
            return cls(graph, dn, kind, dist.loc, dist.scale, **kw)
        if kind == _lib.NOISE_UNIFORM:
            return cls(graph, dn, kind, dist.low, dist.high, **kw)
        return cls(graph, dn, kind, dist.probs, None, **kw)

    @property
    def shape(self):
        return torch.Size([self.graph.number_of_edges(), self.dn])

    def unsqueeze(self, dim):   # stag/zoo/gat.py:118 does edge_weight.unsqueeze(-1)
        return self

    def spec(self):
        s = _lib.NoiseSpec()
        s.kind, s.param_mode = self.kind, self.param_mode
        s.p0, s.p1 = _lib.ptr(self.p0), _lib.ptr(self.p1)
        s.p0_scalar, s.p1_scalar = self.p0_scalar, self.p1_scalar
        s.relu, s.in_norm, s.deriv = int(self.relu), int(self.in_norm), int(self.deriv)
        s.seed, s.offset, s.pos_base = self.seed, self.offset, self.pos_base
        s.chunk_base = self.chunk_base
        s.p1_log = int(self.p1_log)
        s.epoch = _lib.ptr(self.epoch)
        return s

    def torch_args(self, in_norm=None, group=0):
        """(noise_ints, noise_u64, noise_floats, p0, p1, epoch) of torch.ops.stag.* — the same fields as spec()."""
        s64 = lambda v: v - (1 << 64) if v >= (1 << 63) else v          # 64-bit pattern in an int64
        return ([self.kind, self.param_mode, int(self.relu), int(self.in_norm if in_norm is None else in_norm),
                 int(self.deriv), group, self.chunk_base, int(self.p1_log)],
                [s64(self.seed), s64(self.offset), self.pos_base], [self.p0_scalar, self.p1_scalar],
                self.p0, self.p1, self.epoch)

    def materialize(self):
        """The [E, Dn] tensor this descriptor stands for.  With live parameters (vi=True, `grad_params`)
        and gradients enabled it is the reparameterised sample itself, w = p0 + p1 * z (Normal) or
        low + (high - low) * u (Uniform) with the standard draw z | u taken from the same counters — so
        whatever is computed from it (the sample-based KL of stag/layers.py:141-143, whose reference form
        differentiates through `rsample`) sends gradients to loc / log_scale."""
        from . import ops
        live = (self.grad_params is not None and torch.is_grad_enabled()
                and any(torch.is_tensor(p) and p.requires_grad for p in self.grad_params))
        if not live:
            return ops.materialize_noise(self.graph, self)
        import copy
        std = copy.copy(self)
        std.grad_params, std.relu, std.in_norm, std.param_mode, std.p1_log = None, False, False, _lib.PARAM_SCALAR, False
        std.p0 = std.p1 = None
        std.p0_scalar, std.p1_scalar = 0.0, 1.0          # N(0, 1) | U[0, 1)
        z = ops.materialize_noise(self.graph, std)
        p0, p1 = (torch.as_tensor(p, dtype=torch.float32, device=z.device) for p in self.grad_params)
        if self.p1_log:
            p1 = p1.exp()
        w = p0 + p1 * z if self.kind == _lib.NOISE_NORMAL else p0 + (p1 - p0) * z
        if self.relu:
            w = w.relu()
        if self.in_norm:
            from .layers import _in_norm
            w = _in_norm(self.graph, w)
        return w

    def __repr__(self):
        names = {2: "Normal", 3: "Uniform", 4: "Bernoulli"}
        return (f"EdgeNoise({names.get(self.kind, self.kind)}, shape={list(self.shape)}, "
                f"seed={self.seed:#x}, offset={self.offset}, relu={self.relu}, in_norm={self.in_norm})")
