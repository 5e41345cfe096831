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
        batched = self._mc_first_layer_batched(graph, feat, n_samples)
        if batched is not None:
            return batched
        return torch.stack([self._forward(graph, feat) for _ in range(n_samples)], 0).mean(0)

    def _mc_first_layer_batched(self, graph, feat, n_samples):
        """The Monte-Carlo loop with the FIRST layer's samples drawn from one pass over the
        gathered rows (its input is the same for every sample; StagLayer.forward_mc), the other
        layers per sample.  Every sample sees exactly the noise the sequential loop would give it
        (sample s, layer l draws at offset base + s * L + l), so the result is unchanged."""
        first = self.layers[0] if len(self.layers) else None
        if n_samples < 2 or torch.is_grad_enabled() or not hasattr(first, "forward_mc"):
            return None
        stoch = [l for l in self.layers if hasattr(l, "forward_mc")]
        gens = {id(l._generator()) for l in stoch}
        if len(gens) != 1 or not all(l.consumes_offset for l in stoch):
            return None
        gen, L = first._generator(), len(stoch)
        base = gen.offset
        graph = graph.local_var()
        h1 = first.forward_mc(graph, feat, n_samples, offset_stride=L)
        if h1 is None:
            gen.offset = base
            return None
        outs = []
        for s in range(n_samples):
            gen.offset = base + s * L + 1
            h = h1[s]
            for layer in self.layers[1:]:
                h = layer(graph, h)
            outs.append(h)
        gen.offset = base + n_samples * L
        return torch.stack(outs, 0).mean(0)

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
        for _ in range(n_samples):
            out = self._forward(graph, feat)
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
