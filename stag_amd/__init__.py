"""stag_amd — MI355X-native stochastic-aggregation message passing.

Drop-in for the hot path of yuanqing-wang/stag: the `stag.layers` /
`stag.distributions` / `stag.zoo` / `stag.models` names are kept, the work between
`StagLayer.forward` and DGL is one fused HIP kernel (stag_amd/csrc).
"""
from . import distributions, function, layers, likelihoods, models, utils, zoo  # noqa: F401
from . import random  # noqa: F401
from .graph import (Graph, add_reverse_edges, add_self_loop, batch, graph, mean_nodes,  # noqa: F401
                    rand_graph, remove_self_loop, sum_nodes)
from .noise import EdgeNoise  # noqa: F401
from .random import manual_seed  # noqa: F401

__version__ = "0.1.0"
