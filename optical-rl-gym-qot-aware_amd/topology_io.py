"""Topology front-end: build a :class:`FrozenTopology` from a link-list text file instead of a pickled graph.

Mirrors the reference's offline preparation (``examples/create_topology.py:96-147``, ``graph_utils.py:89-116``,
``utils.py:94-117``): nodes are numbered 1..N, every line ``a b length_km`` is an edge whose ``index == id`` is its
line order, the k shortest simple paths per node pair (``networkx.shortest_simple_paths``, weight = length) get the
most efficient modulation whose reach covers the path.  Needs ``networkx`` (host-side preparation only).
"""
from __future__ import annotations

from itertools import islice

from .topology import FrozenTopology

# create_topology.py:47-93 (name, maximum_length km, spectral_efficiency)
DEFAULT_MODULATIONS = (("BPSK", 100_000, 1), ("QPSK", 2_000, 2), ("8QAM", 1_000, 3), ("16QAM", 500, 4),
                       ("32QAM", 250, 5), ("64QAM", 125, 6))


def read_txt_file(path):
    """``graph_utils.py:89-116`` -> (num_nodes, [(a, b, index, length)])."""
    with open(path) as f:
        lines = [ln for ln in f if not ln.startswith("#")]
    num_nodes = int(lines[0])
    edges = []
    for ln in lines[2:]:
        if len(ln) > 1:
            a, b, length = ln.replace("\n", "").split(" ")[:3]
            edges.append((a, b, len(edges), int(length)))
    return num_nodes, edges


def best_modulation(length, modulations):
    """``utils.py:105-117``: the most spectrally efficient format whose reach covers ``length``."""
    for name, max_len, se in sorted(modulations, key=lambda m: m[2], reverse=True):
        if length <= max_len:
            return name, max_len, se
    raise ValueError(f"It was not possible to find a suitable MF for a path with {length} km")


def topology_from_txt(path, name, k_paths=5, modulations=DEFAULT_MODULATIONS) -> FrozenTopology:
    import networkx as nx

    num_nodes, edge_list = read_txt_file(path)
    g = nx.Graph()
    for i in range(1, num_nodes + 1):
        g.add_node(str(i), name=str(i))
    for a, b, idx, length in edge_list:
        g.add_edge(a, b, id=idx, index=idx, weight=1, length=length)
    nodes = [str(n) for n in g.nodes()]
    pair_paths, idp = {}, 0
    for i, a in enumerate(nodes):
        for j, b in enumerate(nodes):
            if i < j:
                plist = []
                for p in islice(nx.shortest_simple_paths(g, a, b, weight="length"), k_paths):
                    length = sum(g[p[q]][p[q + 1]]["length"] for q in range(len(p) - 1))
                    plist.append((idp, len(p) - 1, float(length), best_modulation(length, modulations)[2], list(p)))
                    idp += 1
                pair_paths[a, b] = plist
    edges = [(a, b, idx, idx, float(length)) for a, b, idx, length in edge_list]
    return FrozenTopology(name, nodes, nodes, k_paths, edges, [list(m) for m in modulations], pair_paths)
