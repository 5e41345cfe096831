"""Synthetic graphs of the BASELINE shapes (there is no network for ogbn-arxiv / PPI /
molhiv).  Deterministic numpy generators; used by bench.py and tests."""
import numpy as np

ARXIV_NODES, ARXIV_EDGES, ARXIV_MAX_IN_DEGREE = 169_343, 1_166_243, 13_000


def arxiv_like(n_nodes=ARXIV_NODES, n_edges=ARXIV_EDGES, max_in_degree=ARXIV_MAX_IN_DEGREE,
               n_hubs=200, sigma=1.0, seed=1):
    """ogbn-arxiv-shaped directed graph: skewed in-degree (log-normal body plus `n_hubs`
    Zipf hubs, the largest with about `max_in_degree` in-edges; mean 6.9, ~10 % of nodes
    with no in-edge at the default size), uniformly random sources, edges in random order.
    Returns (src, dst) int64 arrays."""
    rng = np.random.default_rng(seed)
    body = np.exp(sigma * rng.standard_normal(n_nodes))
    body /= body.sum()
    k = min(n_hubs, n_nodes)
    hub = np.zeros(n_nodes)
    hub[:k] = 1.0 / (np.arange(k) + 1.0)
    hub_sum = hub.sum()
    hub /= hub_sum
    hub_mass = min(0.5, max_in_degree * hub_sum / n_edges)
    p = ((1.0 - hub_mass) * body + hub_mass * hub)[rng.permutation(n_nodes)]
    in_deg = rng.multinomial(n_edges, p)
    dst = np.repeat(np.arange(n_nodes, dtype=np.int64), in_deg)
    src = rng.integers(0, n_nodes, n_edges, dtype=np.int64)
    order = rng.permutation(n_edges)
    return src[order], dst[order]


def with_self_loops_and_reverse(src, dst, n_nodes):
    """The reference script's preprocessing (scripts/arxiv_mle/gcn/run.py:53-55):
    remove_self_loop -> add_self_loop -> add_reverse_edges."""
    keep = src != dst
    loop = np.arange(n_nodes, dtype=src.dtype)
    s = np.concatenate([src[keep], loop])
    d = np.concatenate([dst[keep], loop])
    return np.concatenate([s, d]), np.concatenate([d, s])


def molecules_like(n_graphs=4096, mean_nodes=26, seed=2):
    """Batch of small sparse graphs (molhiv-like: ~26 atoms, degree 1-4, both directions).
    Returns (src, dst, batch_num_nodes)."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(rng.poisson(mean_nodes, n_graphs), 2, None)
    srcs, dsts, off = [], [], 0
    for n in sizes:
        parent = np.array([rng.integers(0, i) for i in range(1, n)], dtype=np.int64)  # random tree
        child = np.arange(1, n, dtype=np.int64)
        extra = max(0, int(0.08 * n))                                                  # a few rings
        a, b = rng.integers(0, n, extra), rng.integers(0, n, extra)
        s = np.concatenate([parent, child, a, b]) + off
        d = np.concatenate([child, parent, b, a]) + off
        srcs.append(s); dsts.append(d)
        off += n
    return np.concatenate(srcs), np.concatenate(dsts), sizes.astype(np.int64)


PPI_NODES, PPI_EDGES, PPI_GRAPHS = 56_944, 818_716, 24


def ppi_like(n_graphs=PPI_GRAPHS, n_nodes=PPI_NODES, n_edges=PPI_EDGES, seed=3):
    """Batch of `n_graphs` PPI-sized graphs as ONE block-diagonal graph (`dgl.batch`,
    scripts/ppi_mle/run.py:12-14): sizes spread 0.25x-1.5x around the mean (PPI: 591-3480 nodes),
    edges in proportion to size, skewed in-degree and uniform sources INSIDE each graph, no edge
    between graphs.  Totals are exact.  Returns (src, dst, batch_num_nodes)."""
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.25, 1.5, n_graphs)
    sizes = np.maximum(2, np.floor(w / w.sum() * n_nodes)).astype(np.int64)
    sizes[-1] += n_nodes - sizes.sum()
    ecnt = np.floor(sizes / sizes.sum() * n_edges).astype(np.int64)
    ecnt[-1] += n_edges - ecnt.sum()
    srcs, dsts, off = [], [], 0
    for b in range(n_graphs):
        s, d = arxiv_like(n_nodes=int(sizes[b]), n_edges=int(ecnt[b]), max_in_degree=max(8, int(sizes[b]) // 6),
                          n_hubs=8, sigma=0.9, seed=seed * 1000 + b)
        srcs.append(s + off)
        dsts.append(d + off)
        off += int(sizes[b])
    return np.concatenate(srcs), np.concatenate(dsts), sizes
