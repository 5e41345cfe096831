"""GatedGCN-style layer (reference: stag/zoo/gated_gcn.py:6-61): A(h) + sum_in w (.) h,
then batch norm, relu, optional residual, dropout.  Not in any BASELINE config."""
import torch
import torch.nn.functional as F

from .. import ops
from ._common import check_edge_weight


class GatedGCN(torch.nn.Module):
    supports_edge_noise = True
    supports_edge_noise_grad = True   # vi=True stays fused (ops._AggregateVI)

    def __init__(self, input_dim, output_dim, dropout=0.0, batch_norm=True, residual=False):
        super().__init__()
        self.in_channels, self.out_channels = input_dim, output_dim
        self.dropout, self.batch_norm = dropout, batch_norm
        self.residual = residual and input_dim == output_dim
        self.A = torch.nn.Linear(input_dim, output_dim, bias=True)
        self.B = torch.nn.Linear(input_dim, output_dim, bias=True)
        self.bn_node_h = torch.nn.BatchNorm1d(output_dim)

    def forward(self, g, h, edge_weight=None):
        h_in = h
        if edge_weight is not None:
            check_edge_weight(g, edge_weight)
            summed = ops.aggregate(g, h, edge_weight, reduce="sum")      # u_mul_e('h', w)
        else:
            summed = ops.aggregate(g, self.B(h), None, reduce="sum")     # copy_u('Bh')
        h = self.A(h) + summed
        if self.batch_norm:
            h = self.bn_node_h(h)
        h = F.relu(h)
        if self.residual:
            h = h_in + h
        return F.dropout(h, self.dropout)

    def __repr__(self):
        return f"{self.__class__.__name__}(in_channels={self.in_channels}, out_channels={self.out_channels})"
