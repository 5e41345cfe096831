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
