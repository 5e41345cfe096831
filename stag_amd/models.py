"""Model wrappers with the reference's interface (stag/models.py:7-146): pure-torch
callers of the stochastic layers — the Monte-Carlo loop, NLL + KL objective — kept so
scripts written against `stag.models` run unchanged on top of the HIP layers."""
from typing import List

import torch

from .likelihoods import CategoricalLikelihood, Likelihood


def nll_contrastive(q_a, graph, feat):
    """Contrastive term for an AmortizedDistribution (stag/models.py:7-25): real edges
    should carry weight 1, uniformly drawn fake edges weight 0."""
    n, E = graph.number_of_nodes(), graph.number_of_edges()
    fake_src = torch.randint(high=n, size=[E], device=feat.device)
    fake_dst = torch.randint(high=n, size=[E], device=feat.device)
    h_fake = q_a.embedding_mlp(torch.cat([feat[fake_src], feat[fake_dst]], dim=-1))
    fake = {k: q_a.parameters_mlp[k](h_fake) for k in q_a.new_parameter_names}
    q_neg = q_a.base_distribution_class(
        **{k[4:] if k.startswith("log_") else k: (v.exp() if k.startswith("log_") else v)
           for k, v in fake.items()})
    nll = (-q_a.log_prob(torch.tensor(1.0, device=feat.device))
           - q_neg.log_prob(torch.tensor(0.0, device=feat.device)))
    return nll.sum(dim=-1).mean()


def _masked_mean(nll, mask):
    """nll[mask].mean() (stag/models.py:73-76).  A boolean mask over the nodes is applied as a weight: the same
    value and gradient, without the boolean index (a nonzero + a size read-back: a host synchronisation per
    Monte-Carlo sample, and not capturable in a hipGraph)."""
    if mask is None:
        return nll.mean()
    if mask.device != nll.device:           # nll[mask] takes a host mask for a device tensor; so does this
        mask = mask.to(nll.device)
    if mask.dtype == torch.bool and mask.dim() == 1 and mask.shape[0] == nll.shape[0] and nll.is_cuda:
        per_row = nll.numel() // max(nll.shape[0], 1)
        m = mask.view(-1, *([1] * (nll.dim() - 1)))
        # (where, not a product: an infinite log-probability outside the mask must not turn the sum into NaN)
        return torch.where(m, nll, torch.zeros((), dtype=nll.dtype, device=nll.device)).sum() / (mask.sum() * per_row)
    return nll[mask].mean()


class StagModel(torch.nn.Module):
    def __init__(self, layers: List[torch.nn.Module], likelihood: Likelihood = None,
                 kl_scaling=1.0):
        super().__init__()
        self.layers = layers
        self.likelihood = CategoricalLikelihood() if likelihood is None else likelihood
        self.kl_scaling = kl_scaling

    def _forward(self, graph, feat):
        graph = graph.local_var()
        for layer in self.layers:
            feat = layer(graph, feat)
        return feat

    def _mc_mean(self, graph, feat, n_samples):
        return torch.stack(list(self._mc_outputs(graph, feat, n_samples)), 0).mean(0)

    def _mc_outputs(self, graph, feat, n_samples):
        """The outputs of the Monte-Carlo loop `for _ in range(n_samples): self._forward(graph, feat)`
        (stag/models.py:45-55 at inference, :67-68 in training), one at a time — a generator, so a caller can read
        per-sample layer state (the KL terms of `loss_terms`) right after each sample, as the loop would.

        Where it can, the FIRST layer's samples come from one pass over the gathered rows (its input is the same
        for every sample; `StagLayer.forward_mc` -> stag_agg_fwd_mc) and only the other layers run per sample.
        Every sample sees exactly the noise the sequential loop would give it: with L offsets consumed per sample
        (one per fused draw, one more per in-kernel attention-dropout mask) sample s draws its first layer at
        base + s * L and its other layers from base + s * L + 1 on.  Under autograd, with the first layer's input
        data and its noise fixed (every `*_mle` script), the batched aggregation needs no backward at all — the dense
        transform differentiates through the S outputs; with a learned (`vi=True`) first layer or an input that carries
        a gradient the forward is still batched and the backward is the loop's per-sample passes (ops._AggregateMC)."""
        plan = self._mc_plan(graph, feat, n_samples)
        if plan is None:
            for _ in range(n_samples):
                yield self._forward(graph, feat)
            return
        gen, L, base, first_used, h1, g = plan
        h1 = h1.unbind(0)       # ONE autograd node for the S slices (S separate selects would each zero-fill [S, N, out])
        first = self.layers[0]
        for s in range(n_samples):
            gen.offset = base + s * L + first_used
            first.mc_select(s)          # the first layer's "last draw" is sample s, as after the loop's s-th pass
            h = h1[s]
            for layer in self.layers[1:]:
                h = layer(g, h)
            if gen.offset != base + (s + 1) * L:
                if s == 0:
                    # a layer consumed offsets it does not report (offsets_per_forward): the batched first layer
                    # drew the later samples at the wrong offsets — discard it and run the plain loop, now and
                    # from here on
                    self._mc_batching_off = True
                    gen.offset = base
                    for _ in range(n_samples):
                        yield self._forward(graph, feat)
                    return
                raise RuntimeError("a layer consumed a different number of noise offsets from one Monte-Carlo "
                                   "sample to the next")
            yield h
        gen.offset = base + n_samples * L

    def _mc_plan(self, graph, feat, n_samples):
        """(generator, offsets per sample, base offset, offsets of the first layer, [S, N, out] of the first layer,
        graph) when the first layer's samples can be batched, else None."""
        first = self.layers[0] if len(self.layers) else None
        if n_samples < 2 or not hasattr(first, "forward_mc") or getattr(self, "_mc_batching_off", False):
            return None
        stoch = [l for l in self.layers if hasattr(l, "forward_mc")]
        gens = {id(l._generator()) for l in stoch}
        if len(gens) != 1 or not all(l.consumes_offset for l in stoch):
            return None
        gen = first._generator()
        L = sum(l.offsets_per_forward() for l in stoch)
        first_used = first.offsets_per_forward()
        if first_used != 1:          # (a first layer that draws more than its noise field is not batched)
            return None
        base = gen.offset
        graph = graph.local_var()
        h1 = first.forward_mc(graph, feat, n_samples, offset_stride=L)
        if h1 is None:
            gen.offset = base
            return None
        return gen, L, base, first_used, h1, graph

    def forward(self, graph, feat, n_samples=1, return_parameters=False):
        """Monte-Carlo average over `n_samples` noisy passes; noise stays on at eval time
        (stag/models.py:45-61)."""
        feat = self._mc_mean(graph, feat, n_samples)
        if return_parameters is True:
            return feat
        return self.likelihood.condition(feat).sample()

    def _regulariser(self):
        reg = 0.0
        for layer in self.layers:
            if layer.vi:
                reg = reg + layer.kl_divergence()
        return reg

    def loss_terms(self, graph, feat, y, mask=None, n_samples=1, kl_scaling=None):
        kl_scaling = self.kl_scaling if kl_scaling is None else kl_scaling
        total_nll = total_reg = 0.0
        for out in self._mc_outputs(graph, feat, n_samples):
            nll = -self.likelihood.log_prob(out, y)
            total_nll = total_nll + _masked_mean(nll, mask)
            total_reg = total_reg + self._regulariser()
        return total_nll / n_samples, total_reg / n_samples * kl_scaling

    def loss(self, graph, feat, y, mask=None, n_samples=1, kl_scaling=None):
        nll, reg = self.loss_terms(graph, feat, y, mask=mask, n_samples=n_samples,
                                   kl_scaling=kl_scaling)
        return nll + reg


class StagModelContrastive(StagModel):
    """Adds the last q_a-bearing layer's contrastive NLL to the regulariser
    (stag/models.py:92-146)."""

    def _forward(self, graph, feat):
        graph = graph.local_var()
        contrastive = 0.0
        for layer in self.layers:
            out = layer(graph, feat)
            contrastive = nll_contrastive(layer.q_a, graph, feat) if hasattr(layer, "q_a") else 0.0
            feat = out
        return feat, contrastive

    def _mc_mean(self, graph, feat, n_samples):
        return torch.stack([self._forward(graph, feat)[0] for _ in range(n_samples)], 0).mean(0)

    def loss_terms(self, graph, feat, y, mask=None, n_samples=1, kl_scaling=None):
        kl_scaling = self.kl_scaling if kl_scaling is None else kl_scaling
        total_nll = total_reg = 0.0
        for _ in range(n_samples):
            out, contrastive = self._forward(graph, feat)
            nll = -self.likelihood.log_prob(out, y)
            total_nll = total_nll + _masked_mean(nll, mask)
            total_reg = total_reg + contrastive + self._regulariser()
        return total_nll / n_samples, total_reg / n_samples * kl_scaling
