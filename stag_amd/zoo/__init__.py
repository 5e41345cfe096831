"""Base conv layers with the reference's names and signatures (stag/zoo/__init__.py):
`forward(graph, feat, edge_weight=None)` where edge_weight is a tensor [E, Dn] or an
EdgeNoise descriptor; the aggregation lines run on the fused HIP kernels."""
from .gcn import GCN
from .graph_sage import GraphSAGE
from .gat import GAT
from .gin import GIN
from .gated_gcn import GatedGCN
