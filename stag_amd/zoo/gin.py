"""GIN layer (reference: stag/zoo/gin.py:4-11 = DGL GINConv with a Linear apply_func):
rst = apply_func((1 + eps) * x_dst + sum_{u->v} w[e] (.) x_u)."""
import torch

from .. import ops
from ._common import check_edge_weight


class GIN(torch.nn.Module):
    supports_edge_noise = True
    supports_edge_noise_grad = True   # vi=True stays fused (ops._AggregateVI)
    supports_edge_noise_mc = True     # an EdgeNoise with n_samples > 1 yields [S, N, out]

    def __init__(self, in_features, out_features, aggregator_type="sum", init_eps=0.0,
                 learn_eps=False, activation=None):
        super().__init__()
        if aggregator_type not in ("sum", "mean"):
            raise KeyError(f"Aggregator type {aggregator_type} not recognized.")
        self._aggregator_type = aggregator_type
        self.apply_func = torch.nn.Linear(in_features, out_features)
        self.activation = activation
        if learn_eps:
            self.eps = torch.nn.Parameter(torch.tensor([float(init_eps)]))
        else:
            self.register_buffer("eps", torch.tensor([float(init_eps)]))

    def forward(self, graph, feat, edge_weight=None):
        if edge_weight is not None:
            check_edge_weight(graph, edge_weight)
        neigh = ops.aggregate(graph, feat, edge_weight, reduce=self._aggregator_type)
        h = torch.addcmul(neigh, feat, 1 + self.eps)       # (1 + eps) * feat + neigh in ONE pass (stag/zoo/gin.py:9)
        if isinstance(self.apply_func, torch.nn.Linear):      # split-K weight gradient (ops.node_linear)
            rst = ops.node_linear(h, self.apply_func.weight.t(), self.apply_func.bias)
        else:
            rst = self.apply_func(h)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst
