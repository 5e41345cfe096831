import torch


class DGLError(Exception):
    """Same role as dgl.base.DGLError in the reference's zoo layers."""


def expand_as_pair(feat, graph=None):
    return feat if isinstance(feat, tuple) else (feat, feat)


def check_edge_weight(graph, edge_weight):
    # stag/zoo/gcn.py:61, graph_sage.py:55, gated_gcn.py:32
    assert edge_weight.shape[0] == graph.number_of_edges()


def degree_scale(graph, which, power):
    """deg.clamp(min=1) ** power as fp32, cached on the graph (stag/zoo/gcn.py:68-70,101-103)."""
    owner = graph._cache_owner()
    cache = owner.__dict__.setdefault("_degree_scales", {})
    key = (which, power)
    if key not in cache:
        deg = graph.out_degrees() if which == "out" else graph.in_degrees()
        cache[key] = torch.pow(deg.float().clamp(min=1), power).contiguous()
    return cache[key]
