"""Observation models on top of the last layer's probabilities
(reference: stag/likelihoods.py:1-38; dense epilogue, no graph work)."""
import abc
import os

import torch


VALIDATE_ON_DEVICE = os.environ.get("STAG_VALIDATE_ARGS", "0") == "1"


def _validate(feat):
    """torch's argument validation reads a reduction of the probabilities back to the host — a synchronisation per
    Monte-Carlo sample and not capturable in a hipGraph: off for device tensors (None = torch's default elsewhere).
    Unlike the reference, invalid probabilities (negative, NaN) on the device therefore do not raise here; set
    `stag_amd.likelihoods.VALIDATE_ON_DEVICE = True` (or STAG_VALIDATE_ARGS=1) to get torch's check back while
    debugging."""
    return (None if VALIDATE_ON_DEVICE else False) if feat.is_cuda else None


class Likelihood(torch.nn.Module, abc.ABC):
    def __init__(self, distribution):
        super().__init__()
        self.distribution = distribution

    @abc.abstractmethod
    def condition(self, feat):
        raise NotImplementedError

    def log_prob(self, feat, y):
        return self.condition(feat).log_prob(y)


class CategoricalLikelihood(Likelihood):
    def __init__(self):
        super().__init__(torch.distributions.Categorical)

    def condition(self, feat):
        return self.distribution(probs=feat, validate_args=_validate(feat))


class BernoulliLikelihood(Likelihood):
    def __init__(self):
        super().__init__(torch.distributions.Bernoulli)

    def condition(self, feat):
        return self.distribution(probs=feat, validate_args=_validate(feat))
