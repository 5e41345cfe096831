"""`stag.distributions` API (reference: stag/distributions.py:6-242), kept so the
layer drops in: `Distribution` protocol, `DeltaDistribution`,
`ParametrizedDistribution` (buffers `loc/scale`, or parameters `loc/log_scale`
when vi=True — the state_dict names are part of the contract) and
`AmortizedDistribution` (per-edge parameters from an MLP on [h_src || h_dst]).

These classes only describe q_a; the draw itself happens in the fused kernel
(stag_amd/noise.py, stag_amd/csrc/noise.hpp) unless the layer needs gradients
through the sample (vi=True) or the distribution has no in-kernel sampler.
"""
from functools import partial
from typing import Callable, Union

import torch
from torch.distributions import constraints

_FORWARDED = ("expand", "rsample", "sample", "log_prob", "cdf", "icdf", "entropy")


def _is_positive(constraint):
    if constraint is constraints.positive:
        return True
    inner = getattr(constraint, "base_constraint", None)
    return inner is constraints.positive


class Distribution(torch.nn.Module):
    """nn.Module wrapper that forwards the torch.distributions protocol to
    `self.base_distribution` (stag/distributions.py:6-48)."""

    @property
    def batch_shape(self):
        return self.base_distribution.batch_shape

    @property
    def mean(self):
        return self.base_distribution.mean

    @property
    def stddev(self):
        return self.base_distribution.stddev

    @property
    def variance(self):
        return self.base_distribution.variance

    def condition(self, *args, **kwargs):
        return self


def _forward(name):
    def method(self, *args, **kwargs):
        return getattr(self.base_distribution, name)(*args, **kwargs)
    method.__name__ = name
    return method


for _name in _FORWARDED:
    setattr(Distribution, _name, _forward(_name))


class DeltaDistribution(Distribution):
    """Point mass: sample() is the stored value (stag/distributions.py:50-91)."""

    def __init__(self, value=0.0):
        super().__init__()
        self.register_buffer("value", torch.as_tensor(value))

    batch_shape = property(lambda self: self.value.shape)
    mean = property(lambda self: self.value)
    stddev = property(lambda self: torch.zeros_like(self.value))
    variance = property(lambda self: torch.zeros_like(self.value))

    def sample(self, *args, **kwargs):
        return self.value

    rsample = sample

    def _unsupported(self, *args, **kwargs):
        raise NotImplementedError

    expand = log_prob = cdf = icdf = entropy = _unsupported


class ParametrizedDistribution(Distribution):
    """A torch distribution whose arguments live in the module: buffers when
    vi=False, nn.Parameters when vi=True with positive-constrained ones stored as
    `log_<name>` (stag/distributions.py:93-144)."""

    def __init__(self, base_distribution: torch.distributions.Distribution, vi: bool = False):
        super().__init__()
        cls = type(base_distribution)
        arg_constraints = base_distribution.arg_constraints   # instance level: a property for Uniform
        names = [n for n in arg_constraints if n != "logits"]
        stored = []
        for name in names:
            value = torch.as_tensor(getattr(base_distribution, name)).detach().clone()
            if vi and _is_positive(arg_constraints[name]):
                key, value = "log_" + name, torch.log(value)
            else:
                key = name
            if vi:
                setattr(self, key, torch.nn.Parameter(value))
            else:
                self.register_buffer(key, value)
            stored.append(key)
        self.new_parameter_names = stored
        self.base_distribution_class = partial(cls, validate_args=False)
        self.distribution_type = cls

    def __repr__(self):
        return repr(self.base_distribution)

    def arguments(self):
        """name -> tensor in the distribution's own parametrisation (log_ undone)."""
        return {k[4:] if k.startswith("log_") else k:
                (getattr(self, k).exp() if k.startswith("log_") else getattr(self, k))
                for k in self.new_parameter_names}

    @property
    def base_distribution(self):
        """The torch distribution over the stored arguments.  With buffers (vi=False) the object is kept while the
        buffers are the same tensors at the same version (`.to()`, `load_state_dict`, an in-place update all change
        one or the other): a launch-bound layer call spends 4 us building it and 15 us re-expanding its device
        scalars for the kernel (noise.EdgeNoise keeps those rows per parameter tensor)."""
        if any(isinstance(getattr(self, k), torch.nn.Parameter) for k in self.new_parameter_names):
            return self.base_distribution_class(**self.arguments())
        key = tuple((id(getattr(self, k)), getattr(self, k)._version) for k in self.new_parameter_names)
        cached = self.__dict__.get("_base_cache")
        if cached is None or cached[0] != key:
            cached = (key, self.base_distribution_class(**self.arguments()))
            self.__dict__["_base_cache"] = cached
        return cached[1]


class AmortizedDistribution(Distribution):
    """Per-edge distribution parameters h_e = act(W [h_src || h_dst]), one Linear head
    per argument (stag/distributions.py:146-242).  `condition(graph, feat)` computes
    them; the dense MLP is torch, the kernel consumes the resulting [E, out_features]
    tensors as per-edge loc/scale."""

    def __init__(self, in_features: int, out_features: int,
                 hidden_features: Union[None, int] = None,
                 activation: Callable = torch.nn.SiLU(),
                 base_distribution_class: type = torch.distributions.Normal,
                 init_like: Union[None, torch.distributions.Distribution, Distribution] = None):
        super().__init__()
        hidden_features = out_features if hidden_features is None else hidden_features
        self.new_parameter_names = [
            ("log_" + n) if _is_positive(c) else n
            for n, c in base_distribution_class.arg_constraints.items()]
        self.embedding_mlp = torch.nn.Sequential(
            torch.nn.Linear(2 * in_features, hidden_features), activation)
        self.parameters_mlp = torch.nn.ModuleDict(
            {k: torch.nn.Linear(hidden_features, out_features) for k in self.new_parameter_names})
        self.base_distribution_class = base_distribution_class
        self.distribution_type = base_distribution_class
        self.out_features = out_features
        self.new_parameters = None
        self._base = None
        if init_like is not None:
            self._init_like(init_like)

    def _init_like(self, init_like):
        if isinstance(init_like, Distribution):
            init_like = init_like.base_distribution
        for key in self.new_parameter_names:
            if key.startswith("log_"):
                target = torch.log(torch.as_tensor(getattr(init_like, key[4:]))).mean()
            else:
                target = torch.as_tensor(getattr(init_like, key)).mean()
            torch.nn.init.constant_(self.parameters_mlp[key].bias, float(target))

    def condition(self, graph, feat):
        """h_e = act(W [h_src || h_dst] + b) (stag/distributions.py:178-183, 225-227), computed as
        act((feat W_src^T)[src] + (feat W_dst^T)[dst] + b): the Linear runs over the N node rows
        and the E-row work is two gathers — no [E, 2 in] concatenation, no E-row GEMM — and the
        gathers' backward is the aggregation kernel, not a scatter-add."""
        lin = self.embedding_mlp[0]
        shard = getattr(graph, "is_shard", False)
        if shard and not (feat.is_cuda and feat.dim() == 2 and isinstance(lin, torch.nn.Linear)):
            raise NotImplementedError("AmortizedDistribution on a node-range shard: device features [n_rows, in] and a "
                                      "Linear embedding")
        if feat.is_cuda and feat.dim() == 2 and isinstance(lin, torch.nn.Linear) and self._narrow(graph, lin):
            # narrow heads — AmortizedDistribution(in, 1), hidden_features = 1 by default: what every
            # scripts/*_rec/run.py builds — as three kernels: both projections of feat in ONE pass over it
            # (w = [W_src^T | W_dst^T], the bias on the destination half), then a thread per edge for
            # SiLU and the heads (ops.node_project / ops.edge_mlp, csrc/amort.hip)
            from . import ops
            k, hid = feat.shape[1], lin.out_features
            w = torch.cat([lin.weight[:, :k].t(), lin.weight[:, k:].t()], 1)
            b = None if lin.bias is None else torch.cat([torch.zeros_like(lin.bias), lin.bias])
            heads = [self.parameters_mlp[n] for n in self.new_parameter_names]
            wh = torch.cat([hd.weight for hd in heads], 0).t()
            bs = [hd.bias for hd in heads]
            bh = (None if all(t is None for t in bs) else
                  torch.cat([t if t is not None else torch.zeros(1, device=feat.device) for t in bs]))
            P = ops.node_project(feat, w, b)
            if shard:
                # node-range shard: the per-edge MLP reads the projected row of every edge's SOURCE, local or not — the
                # exchange carries the projected rows ([N, 2 hidden]: 8 bytes per node at hidden = 1, not the features)
                # with the halo lists of the aggregation; the edges then index the buffer (GraphShard.edge_endpoints)
                P = graph.halo_gather(P)
            outs = ops.edge_mlp(graph, P, wh, bh)
            self.new_parameters = dict(zip(self.new_parameter_names, outs))
            self._base = None
            return self
        if feat.is_cuda and feat.dim() == 2 and isinstance(lin, torch.nn.Linear):
            from . import ops
            k = feat.shape[1]
            # the Linear's bias joins the destination half on the N node rows, not the E edge rows
            p_src = ops.node_linear(feat, lin.weight[:, :k].t())
            if shard:       # the source half travels: [n_rows, hidden] -> [n_buf, hidden]
                p_src = graph.halo_gather(p_src)
            h = (ops.gather_rows(graph, p_src, "src")
                 + ops.gather_rows(graph, ops.node_linear(feat, lin.weight[:, k:].t(), lin.bias), "dst"))
            for mod in list(self.embedding_mlp)[1:]:
                h = mod(h)
            heads = [self.parameters_mlp[n] for n in self.new_parameter_names]
            if sum(hd.out_features for hd in heads) <= 64:
                # narrow heads ([E, 1] parameters): ONE product over the E rows of h, then split
                y = ops.node_linear(h, torch.cat([hd.weight for hd in heads], 0).t(),
                                    torch.cat([hd.bias for hd in heads], 0))
                outs = torch.split(y, [hd.out_features for hd in heads], dim=1)
            else:   # wide heads stay separate: their outputs are consumed as contiguous [E, D] rows
                outs = [ops.node_linear(h, hd.weight.t(), hd.bias) for hd in heads]
            self.new_parameters = dict(zip(self.new_parameter_names, outs))
            self._base = None
            return self
        src, dst = graph.edges()
        h = self.embedding_mlp(torch.cat([feat[src], feat[dst]], dim=-1))
        self.new_parameters = {k: self.parameters_mlp[k](h) for k in self.new_parameter_names}
        self._base = None
        return self

    def _narrow(self, graph, lin):
        """The three-kernel form applies: Sequential(Linear, SiLU), hidden <= 4, every head one column wide,
        at most 4 heads."""
        from . import ops
        mods = list(self.embedding_mlp)
        heads = [self.parameters_mlp[n] for n in self.new_parameter_names]
        return (len(mods) == 2 and type(mods[1]) is torch.nn.SiLU and lin.out_features <= ops.NARROW_MAX_HIDDEN
                and len(heads) <= ops.NARROW_MAX_PAR
                and all(isinstance(hd, torch.nn.Linear) and hd.out_features == 1 for hd in heads)
                and hasattr(graph, "_src"))

    def arguments(self):
        if self.new_parameters is None:
            raise RuntimeError("AmortizedDistribution.condition(graph, feat) has not been called")
        return {k[4:] if k.startswith("log_") else k:
                (v.exp() if k.startswith("log_") else v) for k, v in self.new_parameters.items()}

    @property
    def base_distribution(self):
        """Built once per condition() (the layer reads it several times per forward: each build
        is an exp over [E, out_features], and torch's argument validation — two reductions over the
        parameters and a host synchronisation — is skipped: the scale is an exp, positive by
        construction)."""
        if self._base is None:
            self._base = self.base_distribution_class(**self.arguments(), validate_args=False)
        return self._base
