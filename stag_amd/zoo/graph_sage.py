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
        if aggregator_type == "lstm":      # DGL SAGEConv: nn.LSTM(in, in, batch_first=True) over a node's mailbox
            self.lstm = torch.nn.LSTM(in_features, in_features, batch_first=True)
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
        if hasattr(self, "lstm"):
            self.lstm.reset_parameters()

    def _lstm_reduce(self, graph, feat_src, edge_weight):
        """DGL's `_lstm_reducer` over `u_mul_e` messages [DGL, from memory: SAGEConv._lstm_reducer runs
        nn.LSTM(batch_first=True) from a zero state over each destination's mailbox and keeps the last hidden state;
        mailboxes are formed by degree bucketing, messages in edge-id order; a node without in-edges keeps zeros].
        The messages are materialised ([E, D]: a kernel-backed row gather, the noise as a tensor), the destinations
        of equal in-degree are batched through the LSTM.  Cost grows with the number of distinct in-degrees."""
        from ..noise import EdgeNoise
        m = ops.gather_rows(graph, feat_src, "src")                     # [E, D] by edge id
        if isinstance(edge_weight, EdgeNoise):
            edge_weight = edge_weight.materialize()
        if edge_weight is not None:
            m = m * (edge_weight if edge_weight.dim() == 2 else edge_weight.unsqueeze(1))
        csr = graph.csr                                                  # destination-major, stable in edge id
        n = csr.n_dst
        out = m.new_zeros((n, feat_src.shape[1]))
        if csr.n_edges == 0:
            return out
        deg = csr.degrees.long()
        start = csr.indptr[:-1].long()
        eid = csr.eid.long() if csr.eid is not None else torch.arange(csr.n_edges, device=m.device)
        for d in torch.unique(deg).tolist():
            if d == 0:
                continue
            rows = torch.nonzero(deg == d).flatten()
            pos = start[rows].unsqueeze(1) + torch.arange(d, device=m.device).unsqueeze(0)      # [B, d] CSR positions
            box = m[eid[pos]]                                                                   # [B, d, D]
            _, (h, _) = self.lstm(box)
            out = out.index_copy(0, rows, h.squeeze(0))
        return out

    def forward(self, graph, feat, edge_weight=None):
        # (an inactive dropout is skipped, not called: the input then reaches the aggregation as the SAME tensor object,
        # which is how a constant input of odd width is recognised and padded once — ops._padded_constant)
        drop = self.feat_drop if (self.training and self.feat_drop.p > 0) else (lambda t: t)
        if isinstance(feat, tuple):
            feat_src, feat_dst = drop(feat[0]), drop(feat[1])
        else:
            feat_src = feat_dst = drop(feat)
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
        else:   # 'lstm' (stag/zoo/graph_sage.py:97-99): composed, not fused — no BASELINE config or script uses it
            h_neigh = self.fc_neigh(self._lstm_reduce(graph, feat_src, edge_weight))
        rst = h_neigh if self._aggre_type == "gcn" else ops.node_linear(h_self, self.fc_self.weight.t(), add=h_neigh)
        if self.bias is not None and not bias_done:
            rst = ops.add_bias(rst, self.bias)
        if self.activation is not None:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst
