"""GraphSAGE layer (reference: stag/zoo/graph_sage.py:7-119; DGL SAGEConv layout:
`fc_self`, `fc_neigh` without bias plus a shared `bias`).  The neighbour term is
aggregate-then-project (`lin_before_mp = False`, :67), so the aggregation width is
the input width; `mean` and `gcn` run on the fused kernel."""
import torch

from .. import ops
from ._common import check_edge_weight


class GraphSAGE(torch.nn.Module):
    supports_edge_noise = True
    supports_edge_noise_grad = True   # vi=True stays fused (ops._AggregateVI)

    @property
    def supports_edge_noise_mc(self):
        """An EdgeNoise with n_samples > 1 yields [S, N, out] (the Monte-Carlo loop batched on this layer) — for the
        fused aggregators, when no per-sample state hides in the layer: the sequential loop would draw a fresh
        feature-dropout mask per sample, and a `norm` module sees one sample at a time."""
        return (self._aggre_type in ("mean", "gcn") and self.norm is None
                and not (self.training and self.feat_drop.p > 0.0))

    def __init__(self, in_features, out_features, activation=None, aggregator_type="mean",
                 feat_drop=0.0, bias=True, norm=None):
        super().__init__()
        if aggregator_type not in ("mean", "gcn", "pool", "lstm"):
            raise KeyError(f"Aggregator type {aggregator_type} not recognized.")
        self._in_src_feats = self._in_dst_feats = in_features
        self._out_feats, self._aggre_type = out_features, aggregator_type
        self.norm, self.activation = norm, activation
        self.feat_drop = torch.nn.Dropout(feat_drop)
        if aggregator_type == "pool":
            self.fc_pool = torch.nn.Linear(in_features, in_features)
        if aggregator_type != "gcn":
            self.fc_self = torch.nn.Linear(in_features, out_features, bias=False)
        self.fc_neigh = torch.nn.Linear(in_features, out_features, bias=False)
        self.bias = torch.nn.Parameter(torch.zeros(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        gain = torch.nn.init.calculate_gain("relu")
        for name in ("fc_pool", "fc_self", "fc_neigh"):
            if hasattr(self, name):
                torch.nn.init.xavier_uniform_(getattr(self, name).weight, gain=gain)

    def forward(self, graph, feat, edge_weight=None):
        if isinstance(feat, tuple):
            feat_src, feat_dst = self.feat_drop(feat[0]), self.feat_drop(feat[1])
        else:
            feat_src = feat_dst = self.feat_drop(feat)
        if edge_weight is not None:
            check_edge_weight(graph, edge_weight)
        h_self = feat_dst
        bias_done = False
        if self._aggre_type == "mean":
            # the shared bias rides in this GEMM's epilogue, the whole neighbour branch in the self branch's (below):
            # two passes over [N, out] less than `fc_self(h) + fc_neigh(agg) + bias`
            h_neigh = ops.node_linear(ops.aggregate(graph, feat_src, edge_weight, reduce="mean"),
                                      self.fc_neigh.weight.t(), self.bias)
            bias_done = self.bias is not None
        elif self._aggre_type == "gcn":
            neigh = ops.aggregate(graph, feat_src, edge_weight, reduce="sum")
            degs = graph.in_degrees().to(feat_dst)
            h_neigh = ops.node_linear((neigh + feat_dst) / (degs.unsqueeze(-1) + 1), self.fc_neigh.weight.t())
        elif self._aggre_type == "pool":     # max reducer: composed, not fused (ops.aggregate_max)
            h_neigh = self.fc_neigh(ops.aggregate_max(graph, torch.relu(self.fc_pool(feat_src)), edge_weight))
        else:
            raise NotImplementedError("'lstm' aggregator is outside the accelerated path")
        rst = h_neigh if self._aggre_type == "gcn" else ops.node_linear(h_self, self.fc_self.weight.t(), add=h_neigh)
        if self.bias is not None and not bias_done:
            rst = ops.add_bias(rst, self.bias)
        if self.activation is not None:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst
