"""Import-level drop-in: `stag_amd.compat.install()` makes `import stag`, `import dgl`, `import dgl.function as fn`
resolve to this package, so a script written against the reference's imports (stag/layers.py:39-113 call shapes,
scripts/*/run.py) runs unchanged on the HIP path — BASELINE north_star: "drops into the existing stag.models ...
wrappers unchanged".

    import stag_amd.compat; stag_amd.compat.install()      # or: python -m stag_amd.compat script.py [args ...]
    import dgl, stag                                       # the names the reference's scripts use

What `dgl` then offers is the surface stag and its scripts touch, nothing more:
    dgl.graph, dgl.rand_graph, dgl.batch, dgl.remove_self_loop, dgl.add_self_loop, dgl.add_reverse_edges,
    dgl.sum_nodes, dgl.mean_nodes, dgl.DGLGraph, dgl.function (copy_u|copy_src, copy_e|copy_edge, u_mul_e, u_add_v; sum, mean, max),
    dgl.nn.GraphConv | SAGEConv | GATConv | GINConv  (the zoo layers: same constructor arguments, `edge_weight=` forward),
    dgl.base.DGLError, dgl.utils.expand_as_pair | check_eq_shape, dgl.dataloading.GraphDataLoader (lists of graphs ->
    dgl.batch), dgl.nn.functional.edge_softmax.
`dgl.data` (the datasets: they download) is out of scope and says so when touched.  A real `dgl` or `stag` that is
already imported is never replaced (install() raises unless force=True)."""
import importlib
import sys
import types

import torch

_INSTALLED = False


class _Missing(types.ModuleType):
    """A namespace of the real DGL this package does not stand in for: touching it explains instead of AttributeError."""

    def __init__(self, name, why):
        super().__init__(name)
        self.__dict__["_why"] = why

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        raise ImportError(f"{self.__name__}.{item}: {self._why}")


class GraphDataLoader:
    """`dgl.dataloading.GraphDataLoader` for a sequence of graphs (scripts/ppi_mle/run.py:12-14: PPIDataset items are
    graphs whose labels live in ndata): batches of `batch_size` graphs through dgl.batch, optionally shuffled."""

    def __init__(self, dataset, batch_size=1, shuffle=False, drop_last=False, generator=None, **_unused):
        self.dataset, self.batch_size, self.shuffle, self.drop_last, self.generator = dataset, int(batch_size), shuffle, drop_last, generator

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        batch = importlib.import_module("stag_amd.graph").batch
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for i in range(0, n, self.batch_size):
            idx = order[i:i + self.batch_size]
            if len(idx) < self.batch_size and self.drop_last:
                return
            items = [self.dataset[j] for j in idx]
            if isinstance(items[0], tuple):             # (graph, label) datasets
                yield (batch([t[0] for t in items]),) + tuple(torch.stack([torch.as_tensor(t[k]) for t in items])
                                                              for k in range(1, len(items[0])))
            else:
                yield batch(items)


def _edge_softmax(graph, e):
    """`dgl.nn.functional.edge_softmax(graph, e)`: softmax of e [E, ...] over the in-edges of every destination
    (stag/zoo/gat.py:120-122 applies it to the noisy logits).  Composed (the fused form is ops.gat_aggregate)."""
    from . import ops
    _, dst = graph.edges()
    shape = e.shape
    e2 = e.reshape(shape[0], -1)
    n = graph.number_of_dst_nodes()
    m = torch.full((n, e2.shape[1]), float("-inf"), dtype=e2.dtype, device=e2.device)
    m = m.scatter_reduce(0, dst.unsqueeze(1).expand(-1, e2.shape[1]), e2.detach(), reduce="amax", include_self=True)
    p = torch.exp(e2 - m[dst])
    den = ops.aggregate(graph, torch.ones(1, e2.shape[1], device=e2.device), p, _broadcast_x=True)
    return (p / ops.gather_rows(graph, den, "dst")).reshape(shape)


def _check_eq_shape(feat):
    src, dst = feat
    if src.shape[1:] != dst.shape[1:]:
        from .zoo._common import DGLError
        raise DGLError("The feature shape of source nodes and destination nodes must match")


def _dgl_module():
    from . import function, zoo
    from .zoo import _common
    G = importlib.import_module("stag_amd.graph")      # (the package attribute `graph` is the dgl.graph constructor)
    dgl = types.ModuleType("dgl")
    dgl.__doc__ = "stag_amd's stand-in for the DGL surface yuanqing-wang/stag touches (stag_amd.compat)"
    dgl.__version__ = "0.0+stag_amd"
    dgl.__stag_amd_compat__ = True
    for name in ("graph", "rand_graph", "batch", "remove_self_loop", "add_self_loop", "add_reverse_edges", "sum_nodes",
                 "mean_nodes"):
        setattr(dgl, name, getattr(G, name))
    dgl.DGLGraph = dgl.DGLHeteroGraph = G.Graph
    fn = types.ModuleType("dgl.function")
    fn.__dict__.update({k: v for k, v in vars(function).items() if not k.startswith("_")})
    nn = types.ModuleType("dgl.nn")
    nn.GraphConv, nn.SAGEConv, nn.GATConv, nn.GINConv = zoo.GCN, zoo.GraphSAGE, zoo.GAT, zoo.GIN
    nnf = types.ModuleType("dgl.nn.functional")
    nnf.edge_softmax = _edge_softmax
    nn.functional = nnf
    nn_pt = types.ModuleType("dgl.nn.pytorch")
    nn_pt.__dict__.update({k: v for k, v in vars(nn).items() if not k.startswith("_")})
    nn.pytorch = nn_pt
    base = types.ModuleType("dgl.base")
    base.DGLError = _common.DGLError
    utils = types.ModuleType("dgl.utils")
    utils.expand_as_pair, utils.check_eq_shape = _common.expand_as_pair, _check_eq_shape
    dataloading = types.ModuleType("dgl.dataloading")
    dataloading.GraphDataLoader = GraphDataLoader
    data = _Missing("dgl.data", "DGL's datasets download from the network and are outside this package's scope "
                                "(SURVEY.md section 2): build the graph with dgl.graph((src, dst), num_nodes=n) and "
                                "put features / labels / masks into g.ndata")
    dgl.function, dgl.nn, dgl.base, dgl.utils, dgl.dataloading, dgl.data = fn, nn, base, utils, dataloading, data
    dgl.DGLError = _common.DGLError
    dgl.expand_as_pair = _common.expand_as_pair
    mods = {"dgl": dgl, "dgl.function": fn, "dgl.nn": nn, "dgl.nn.pytorch": nn_pt, "dgl.nn.functional": nnf,
            "dgl.base": base, "dgl.utils": utils, "dgl.dataloading": dataloading, "dgl.data": data}
    return mods


def install(force=False):
    """Alias `stag` (and its submodules) and `dgl` to this package in sys.modules.  Idempotent.  Raises ImportError when
    a real `dgl` or `stag` is already imported (force=True replaces them)."""
    global _INSTALLED
    import stag_amd
    for name in ("stag", "dgl"):
        have = sys.modules.get(name)
        if have is not None and have is not stag_amd and not getattr(have, "__stag_amd_compat__", False) and not force:
            raise ImportError(f"a real `{name}` is already imported ({getattr(have, '__file__', '?')}); "
                              f"stag_amd.compat.install(force=True) replaces it")
    sys.modules["stag"] = stag_amd
    for sub in ("layers", "distributions", "zoo", "models", "likelihoods", "utils", "function", "random"):
        sys.modules[f"stag.{sub}"] = importlib.import_module(f"stag_amd.{sub}")
    for sub in ("gcn", "graph_sage", "gat", "gin", "gated_gcn"):
        sys.modules[f"stag.zoo.{sub}"] = importlib.import_module(f"stag_amd.zoo.{sub}")
    sys.modules.update(_dgl_module())
    _INSTALLED = True
    return stag_amd


def installed():
    return _INSTALLED


def uninstall():
    """Remove the aliases again (tests)."""
    global _INSTALLED
    import stag_amd
    ours = getattr(sys.modules.get("dgl"), "__stag_amd_compat__", False)
    for name in list(sys.modules):
        m = sys.modules[name]
        if (name == "stag" or name.startswith("stag.")) and (m is stag_amd or getattr(m, "__name__", "").startswith("stag_amd")):
            del sys.modules[name]
        elif (name == "dgl" or name.startswith("dgl.")) and ours:
            del sys.modules[name]
    _INSTALLED = False


if __name__ == "__main__":      # python -m stag_amd.compat script.py [args ...]: run a script written against stag / dgl
    import runpy
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m stag_amd.compat script.py [args ...]")
    install()
    sys.argv = sys.argv[1:]
    runpy.run_path(sys.argv[0], run_name="__main__")
