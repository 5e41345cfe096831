"""GCN layer: DGL GraphConv semantics with an edge weight (reference: stag/zoo/gcn.py).

    x'  = x * outdeg^-1/2                      (norm='both', zoo/gcn.py:67-75)
    agg = sum_{u->v} w[e] (.) x'[u]            (zoo/gcn.py:94-96)  -- always BEFORE
    rst = agg @ W                              (zoo/gcn.py:97-98)     the matmul (:85)
    rst = rst * indeg^-1/2 + bias ; activation (zoo/gcn.py:100-114)

The two degree scalings, the weight draw and the segmented sum are one kernel
launch (`ops.aggregate`); the row scaling by indeg^-1/2 commutes with `@ W`, so it
is applied in the kernel's epilogue instead of after the matmul.
"""
import torch

from .. import ops
from ._common import DGLError, check_edge_weight, degree_scale, expand_as_pair


class GCN(torch.nn.Module):
    supports_edge_noise = True
    supports_edge_noise_grad = True   # vi=True stays fused (ops._AggregateVI)
    supports_edge_noise_mc = True     # an EdgeNoise with n_samples > 1 yields [S, N, out]

    def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None,
                 allow_zero_in_degree=False):
        super().__init__()
        if norm not in ("none", "both", "right", "left"):
            raise DGLError(f'Invalid norm value "{norm}"')
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        self.weight = torch.nn.Parameter(torch.empty(in_feats, out_feats)) if weight else None
        self.bias = torch.nn.Parameter(torch.empty(out_feats)) if bias else None
        self._activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight is not None:
            torch.nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            torch.nn.init.zeros_(self.bias)

    def forward(self, graph, feat, weight=None, edge_weight=None):
        if edge_weight is not None:
            check_edge_weight(graph, edge_weight)
        feat_src, _ = expand_as_pair(feat, graph)
        src_scale = dst_scale = None
        if self._norm in ("left", "both"):
            src_scale = degree_scale(graph, "out", -0.5 if self._norm == "both" else -1.0)
        if self._norm in ("right", "both"):
            dst_scale = degree_scale(graph, "in", -0.5 if self._norm == "both" else -1.0)
        if weight is not None:
            if self.weight is not None:
                raise DGLError("External weight is provided while at the same time the module has "
                               "defined its own weight parameter. Please create the module with "
                               "flag weight=False.")
        else:
            weight = self.weight
        lead = feat_src.shape
        rst = ops.aggregate(graph, feat_src.reshape(lead[0], -1), edge_weight, reduce="sum",
                            src_scale=src_scale, dst_scale=dst_scale)
        rst = rst.reshape(lead if rst.dim() == 2 else (rst.shape[0],) + tuple(lead))   # [S, ...]: MC samples
        if weight is not None and self.bias is not None and rst.dim() == 2:
            rst = ops.node_linear(rst, weight, self.bias)        # bias in the GEMM's epilogue
        else:
            if weight is not None:
                rst = ops.node_linear(rst, weight)
            if self.bias is not None:
                rst = rst + self.bias
        if self._activation is not None:
            rst = self._activation(rst)
        return rst

    def extra_repr(self):
        return f"in={self._in_feats}, out={self._out_feats}, normalization={self._norm}"
