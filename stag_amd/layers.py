"""`stag.layers` API on the MI355X path (reference: stag/layers.py:8-178).

`StagLayer(base_layer, q_a, p_a, norm, relu, vi).forward(graph, feat)` keeps the
reference's contract, but where the reference samples an [E, Dn] tensor and
hands it down (stag/layers.py:96-113), this layer hands down an `EdgeNoise`
descriptor and the base layer's aggregation kernel draws the weights itself.
The tensor only appears when (a) gradients must flow into q_a (vi=True: the
reparameterised sample w = loc + scale * z is formed with z drawn by the HIP
kernel), (b) q_a has no in-kernel sampler, or (c) the base layer cannot take a
descriptor; in all three the aggregation still runs on the explicit-weight
kernel, never on a CPU or eager-PyTorch fallback.
"""
from typing import Union

import torch

from . import _lib
from . import function as fn
from . import graph as _graph
from . import random as _random
from .distributions import Distribution, ParametrizedDistribution
from .noise import EdgeNoise, fusable


def _in_norm(graph, edge_weight_sample):
    """Rescale weights so each destination's incoming weights sum to its in-degree,
    per channel (stag/layers.py:8-36): s = indeg / sum_in(w) where the sum is
    non-zero, else 1.  Tensor form, used when the weights are materialised."""
    graph = graph.local_var()
    graph.edata["h_a"] = edge_weight_sample
    graph.update_all(fn.copy_e("h_a", "m_a"), fn.sum("m_a", "h_a"))
    current = graph.ndata["h_a"]
    desired = graph.in_degrees().unsqueeze(-1).to(current.dtype)
    scaling = torch.where(current != 0.0, desired / current, torch.ones_like(current))
    from . import ops
    return edge_weight_sample * ops.gather_rows(graph, scaling, "dst")


class StagLayer(torch.nn.Module):
    """Make a graph conv layer stochastic: multiply every message by a freshly drawn
    per-edge, per-channel weight ~ q_a each forward pass."""

    def __init__(
        self,
        base_layer: torch.nn.Module,
        q_a: Union[Distribution, torch.distributions.Distribution] = torch.distributions.Normal(1.0, 1.0),
        p_a: Union[None, Distribution, torch.distributions.Distribution] = torch.distributions.Normal(1.0, 1.0),
        norm: bool = False,
        relu: bool = False,
        vi: bool = False,
        generator: Union[None, _random.NoiseGenerator] = None,
    ) -> None:
        super().__init__()
        self.base_layer = base_layer
        if isinstance(q_a, torch.distributions.Distribution):
            q_a = ParametrizedDistribution(q_a, vi=vi)
        if isinstance(p_a, torch.distributions.MixtureSameFamily):
            p_a.base_distribution = p_a
        elif isinstance(p_a, torch.distributions.Distribution):
            p_a = ParametrizedDistribution(p_a, vi=vi)
        elif p_a is None:
            p_a = ParametrizedDistribution(q_a.base_distribution, vi=vi)
        self.add_module("q_a", q_a)
        self.p_a = p_a
        self.norm, self.relu, self.vi = norm, relu, vi
        self.generator = generator
        self._edge_weight_handle = None

    # ---- the [E, Dn] weight of the last forward (stag/layers.py:107) ----------------
    @property
    def _edge_weight_sample(self):
        h = self._edge_weight_handle
        if isinstance(h, EdgeNoise):
            h = h.materialize()
            self._edge_weight_handle = h
        return h

    @_edge_weight_sample.setter
    def _edge_weight_sample(self, value):
        self._edge_weight_handle = value

    def _generator(self):
        return self.generator if self.generator is not None else _random.default_generator

    def _sample_dimension(self, feat):
        # GAT draws one weight per head (stag/zoo/gat.py:11), everything else per channel
        return getattr(self.base_layer, "sample_dimension", feat.shape[-1])

    def forward(self, graph, feat):
        graph = graph.local_var()
        self._set_kl_share(graph)
        self.q_a.condition(graph, feat)
        dn = self._sample_dimension(feat)
        w = self.rsample_noise(graph, dn)
        if not isinstance(w, EdgeNoise):
            if self.relu:
                w = w.relu()
            if self.norm:
                w = _in_norm(graph, w)
        self._edge_weight_handle = w
        if hasattr(self.base_layer, "noise_generator"):      # a base layer that draws itself (GAT's attention dropout)
            self.base_layer.noise_generator = self._generator()
        return self.base_layer.forward(graph=graph, feat=feat, edge_weight=w)

    @property
    def consumes_offset(self):
        """True if a forward pass takes exactly one offset from the generator (a fusable q_a)."""
        try:
            return fusable(self.q_a.base_distribution)
        except Exception:      # an AmortizedDistribution that was never conditioned
            return False

    def offsets_per_forward(self):
        """Offsets of the generator one forward pass consumes: one per fused draw, plus what the base layer draws
        itself (GAT's in-kernel attention dropout)."""
        extra = getattr(self.base_layer, "extra_offsets_per_forward", None)
        return (1 if self.consumes_offset else 0) + (extra() if callable(extra) else 0)

    def _set_kl_share(self, graph):
        """On a node-range shard the KL term is this rank's SHARE (kl_divergence): (E_local / E_global, 1 / world);
        None on a whole graph.  Set by every call that sees a graph, so the value never outlives it."""
        self._kl_share = ((graph.number_of_edges() / max(graph.n_edges_global, 1), 1.0 / graph.world)
                          if getattr(graph, "is_shard", False) else None)

    def forward_mc(self, graph, feat, n_samples, offset_stride=1):
        """n_samples forward passes on the SAME input from one pass over the gathered rows:
        [n_samples, N, out], sample s drawn at this call's offset + s * offset_stride (what the
        sequential Monte-Carlo loop of stag/models.py:45-55 would use when every pass consumes
        offset_stride offsets).  None when the layer cannot batch (the caller then loops)."""
        if not getattr(self.base_layer, "supports_edge_noise_mc", False):
            return None
        graph = graph.local_var()
        self._set_kl_share(graph)      # (every entry that conditions on a graph: a stale share would scale the KL term)
        self.q_a.condition(graph, feat)
        dist = self.q_a.base_distribution
        if not fusable(dist):
            return None
        reparam = type(dist) in (torch.distributions.Normal, torch.distributions.Uniform)
        live = torch.is_grad_enabled() and self.vi
        if live and not (reparam and getattr(self.base_layer, "supports_edge_noise_grad", False)):
            return None          # (a learned distribution without a reparameterised in-kernel draw: the loop)
        if torch.is_grad_enabled() and self.norm and (live or feat.requires_grad):
            return None          # (in-norm's factor is differentiated by ops._AggregateVI / _Aggregate, per sample)
        dn = self._sample_dimension(feat)
        gen = self._generator()
        # under autograd the batched forward shares the gathers and the backward is the loop's per-sample passes
        # (ops._AggregateMC); with fixed noise on an input that is data (every `*_mle` script) there is no backward at all
        w = self._descriptor(graph, dn, dist, relu=self.relu, in_norm=self.norm, differentiable=live, seed=gen.seed,
                             offset=gen.offset, epoch=gen.device_epoch)
        if w.param_mode > _lib.PARAM_PER_CHANNEL:
            return None
        gen.next_offset()
        w.n_samples, w.offset_stride = int(n_samples), int(offset_stride)
        self._edge_weight_handle = None      # no single [E, Dn] sample stands for this call: mc_select(s) names one
        self._mc_noise = w
        return self.base_layer.forward(graph=graph, feat=feat, edge_weight=w)

    def mc_select(self, s):
        """After forward_mc: make sample `s` the layer's current draw — what `_edge_weight_sample` (stag/layers.py:107)
        and the sample-based KL fallback (:141-143) read, as they would after the s-th pass of the sequential loop."""
        import copy
        w = getattr(self, "_mc_noise", None)
        if w is None:
            return
        one = copy.copy(w)
        one.offset = w.offset + int(s) * w.offset_stride
        one.n_samples = 1
        self._edge_weight_handle = one

    def _descriptor(self, graph, dn, dist, **kw):
        """EdgeNoise of q_a.  An AmortizedDistribution's Normal hands over its heads' outputs as they are —
        `loc` and `log_scale`, [E, 1 | Dn] — and the kernels exponentiate the log-scale where they load it
        (stag_noise_spec.p1_log): the [E, Dn] exp pass of stag/distributions.py:235-242 and the tensor autograd
        would keep for it do not exist, and the gradient comes back w.r.t. `log_scale` directly."""
        q = self.q_a
        params = getattr(q, "new_parameters", None)
        if (isinstance(dist, torch.distributions.Normal) and isinstance(params, dict)
                and set(params) == {"loc", "log_scale"}):
            return EdgeNoise(graph, dn, _lib.NOISE_NORMAL, params["loc"], params["log_scale"], p1_log=True, **kw)
        return EdgeNoise.from_distribution(graph, dn, dist, **kw)

    def rsample_noise(self, graph, sample_dimension):
        """Edge weights of shape [E, sample_dimension] (stag/layers.py:115-129): an
        EdgeNoise descriptor when the draw can be fused into the aggregation (relu and
        in-norm ride along in the descriptor), else a tensor."""
        E = graph.number_of_edges()
        dist = self.q_a.base_distribution
        gen = self._generator()
        fused_ok = fusable(dist) and getattr(self.base_layer, "supports_edge_noise", False)
        if fused_ok and not self.vi:
            return self._descriptor(
                graph, sample_dimension, dist, relu=self.relu, in_norm=self.norm,
                seed=gen.seed, offset=gen.next_offset(), epoch=gen.device_epoch)
        reparam = type(dist) in (torch.distributions.Normal, torch.distributions.Uniform)
        if (fused_ok and self.vi and reparam
                and getattr(self.base_layer, "supports_edge_noise_grad", False)):
            # vi=True on the fused path: the descriptor keeps the live loc / scale tensors and
            # ops.aggregate returns their gradients by regenerating the noise in the backward
            # (in-norm included: its factor is differentiated from two [N, D] tensors, ops._AggregateVI)
            return self._descriptor(
                graph, sample_dimension, dist, relu=self.relu, in_norm=self.norm, differentiable=True,
                seed=gen.seed, offset=gen.next_offset(), epoch=gen.device_epoch)
        if fusable(dist) and self.vi and type(dist) in (torch.distributions.Normal,
                                                       torch.distributions.Uniform):
            # reparameterised draw: standard noise from the HIP stream, affine map in
            # torch so autograd reaches loc / log_scale (rsample, stag/layers.py:123-124)
            std = (EdgeNoise(graph, sample_dimension, _lib.NOISE_NORMAL, 0.0, 1.0,
                             seed=gen.seed, offset=gen.next_offset(), epoch=gen.device_epoch)
                   if isinstance(dist, torch.distributions.Normal) else
                   EdgeNoise(graph, sample_dimension, _lib.NOISE_UNIFORM, 0.0, 1.0,
                             seed=gen.seed, offset=gen.next_offset(), epoch=gen.device_epoch))
            z = std.materialize()
            if isinstance(dist, torch.distributions.Normal):
                return dist.loc + dist.scale * z
            return dist.low + (dist.high - dist.low) * z
        if fusable(dist):   # base layer that cannot take a descriptor: draw, then hand a tensor
            with torch.no_grad():
                return EdgeNoise.from_distribution(graph, sample_dimension, dist,
                                                   seed=gen.seed, offset=gen.next_offset(), epoch=gen.device_epoch).materialize()
        expanded = self.q_a.expand([E, sample_dimension])
        if self.vi:
            return expanded.rsample()
        with torch.no_grad():
            return expanded.sample()

    def kl_divergence(self):
        """KL(q_a || p_a) averaged over parameter entries; sample-based estimate when
        torch has no closed form for the pair (stag/layers.py:132-145)."""
        if not self.vi:
            return 0.0
        kl = self._kl_unweighted()
        w = self._kl_weight()
        return kl if w == 1.0 else kl * w

    def _kl_weight(self):
        """1 on a whole graph.  On a node-range shard every rank returns its SHARE, so that the ranks' values (and
        gradients) SUM to the whole graph's: per-edge parameters (an AmortizedDistribution's heads, [E_local, .]) —
        and the sample-based estimate, a mean over this rank's edges — count E_local / E_global, replicated
        parameters 1 / world."""
        share = getattr(self, "_kl_share", None)
        if share is None:
            return 1.0
        per_edge = isinstance(getattr(self.q_a, "new_parameters", None), dict)
        return share[0] if (per_edge or getattr(self, "_kl_sampled", False)) else share[1]

    def _kl_unweighted(self):
        self._kl_sampled = False
        fused = self._kl_normal_fused()
        if fused is not None:
            return fused
        try:
            return torch.distributions.kl_divergence(
                self.q_a.base_distribution, self.p_a.base_distribution).mean()
        except Exception:   # the reference falls back on ANY failure (stag/layers.py:141)
            self._kl_sampled = True
            w = self._edge_weight_sample
            return (self.q_a.log_prob(w).sum(dim=-1).mean()
                    - self.p_a.log_prob(w).sum(dim=-1).mean())


    def _kl_normal_fused(self):
        """KL of a Normal q_a that keeps (loc, log_scale) — an amortised one's heads [E, out], or a vi=True
        ParametrizedDistribution's parameters — against a Normal prior with one-element parameters: one pass
        forward, one backward (ops.normal_kl_mean); torch's closed form is ~25 elementwise launches (passes over
        the [E, out] heads, or as many one-element launches: 100 us of a 790 us layer step).  None when the pair is
        anything else."""
        q = self.q_a
        params = getattr(q, "new_parameters", None)
        if (isinstance(params, dict) and set(params) == {"loc", "log_scale"}
                and getattr(q, "base_distribution_class", None) is torch.distributions.Normal):
            loc, ls = params["loc"], params["log_scale"]              # amortised heads, [E, out]
        elif (isinstance(q, ParametrizedDistribution) and getattr(q, "distribution_type", None) is torch.distributions.Normal
              and list(q.new_parameter_names) == ["loc", "log_scale"]):
            loc, ls = q.loc, q.log_scale                              # learned scalars / rows (vi=True): a dozen
        else:                                                         # one-element launches otherwise
            return None
        if not (torch.is_tensor(loc) and loc.is_cuda and loc.shape == ls.shape and loc.numel() > 0):
            return None
        try:
            prior = self.p_a.base_distribution
        except Exception:
            return None
        if type(prior) is not torch.distributions.Normal or prior.loc.numel() != 1 or prior.scale.numel() != 1:
            return None
        if not prior.loc.is_cuda:
            return None
        from . import ops as _ops
        return _ops.normal_kl_mean(loc, ls, prior.loc, prior.scale)


class FeatOnlyLayer(torch.nn.Module):
    """Apply a dense module, ignore the graph (stag/layers.py:147-154)."""
    vi = False

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, graph, feat):
        return self.layer(feat)


class SumNodes(torch.nn.Module):
    """Per-graph sum readout of a batched graph (stag/layers.py:156-166)."""
    vi = False

    def __init__(self, name="to_sum"):
        super().__init__()
        self.name = name

    def forward(self, graph, feat):
        graph = graph.local_var()
        graph.ndata[self.name] = feat
        return _graph.sum_nodes(graph, self.name)


class MeanNodes(torch.nn.Module):
    """Per-graph mean readout of a batched graph (stag/layers.py:168-178)."""
    vi = False

    def __init__(self, name="to_mean"):
        super().__init__()
        self.name = name

    def forward(self, graph, feat):
        graph = graph.local_var()
        graph.ndata[self.name] = feat
        return _graph.mean_nodes(graph, self.name)
