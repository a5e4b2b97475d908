"""Frozen topology tables.

The reference keeps the network as a pickled ``networkx.Graph`` whose ``graph`` dict holds the
k-shortest-path objects (``create_topology.py:96-147``) and looks links up through
``topology[a][b]["index"]`` on every access (``rmsa_env.py:479-483,725-729``).  Here the same
information is frozen ONCE, at environment construction, into flat CSR tables that are copied to
the GPU: per ordered node pair the k path records, per path its hop count, spectral efficiency,
length and link-index list.

``FrozenTopology.from_graph`` accepts the reference's own graph objects (duck-typed: anything with
``.graph[...]``, ``.nodes()``, ``.edges()`` and ``g[a][b]``), ``FrozenTopology.from_json`` loads
the JSON form written by ``tests/golden/make_golden.py`` / :meth:`to_json`.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

MAX_HOPS = 14          # link ids of one path are packed in a 16-byte record on the device
MAX_NODES = 64         # src/dst sampling is one wavefront ballot
MAX_LINKS = 255        # link index is a byte in the device path record


@dataclass
class Modulation:
    """Mirror of ``optical_rl_gym/utils.py:14-24``."""
    name: str
    maximum_length: Union[int, float]
    spectral_efficiency: int
    minimum_osnr: Optional[float] = field(default=None)
    inband_xt: Optional[float] = field(default=None)


@dataclass
class Path:
    """Mirror of ``optical_rl_gym/utils.py:27-36`` (+ ``gid``: row in the frozen path table)."""
    path_id: int
    node_list: Tuple[str, ...]
    hops: int
    length: Union[int, float]
    idp: Optional[int] = field(default=0)
    best_modulation: Optional[Modulation] = field(default=None)
    current_modulation: Optional[Modulation] = field(default=None)
    gid: int = field(default=-1)


@dataclass(repr=False)
class Service:
    """Mirror of ``optical_rl_gym/utils.py:56-80``."""
    service_id: int
    source: str
    source_id: int
    destination: Optional[str] = field(default=None)
    destination_id: Optional[int] = field(default=None)
    arrival_time: Optional[float] = field(default=None)
    holding_time: Optional[float] = field(default=None)
    bit_rate: Optional[float] = field(default=None)
    path: Optional[Path] = field(default=None)
    best_modulation: Optional[Modulation] = field(default=None)
    service_class: Optional[int] = field(default=None)
    number_slots: Optional[int] = field(default=None)
    channels: Optional[list] = field(default=None)
    core: Optional[int] = field(default=None)
    launch_power: Optional[float] = field(default=None)
    accepted: bool = field(default=False)
    virtual_layer: bool = field(default=False)

    def __str__(self):
        msg = "{"
        msg += "" if self.bit_rate is None else f"br: {self.bit_rate}, "
        msg += "" if self.service_class is None else f"cl: {self.service_class}, "
        return f"Serv. {self.service_id} ({self.source} -> {self.destination})" + msg


class _Adjacency(dict):
    """``topology[a][b]`` -> edge attribute dict."""


class TopologyView:
    """The small part of the ``networkx.Graph`` surface that envs and heuristics touch
    (``topology.graph[...]``, ``topology[a][b]["index"]``, ``nodes()``, ``edges()``,
    ``number_of_nodes()``, ``number_of_edges()``)."""

    def __init__(self, nodes: Sequence[str], edges: Sequence[Tuple[str, str, dict]], graph: dict):
        self.graph = graph
        self._nodes = list(nodes)
        self._edges = [(a, b) for a, b, _ in edges]
        self._adj: Dict[str, _Adjacency] = {n: _Adjacency() for n in self._nodes}
        for a, b, attrs in edges:
            self._adj[a][b] = attrs
            self._adj[b][a] = attrs

    def __getitem__(self, n):
        return self._adj[n]

    def nodes(self):
        return list(self._nodes)

    def edges(self):
        return list(self._edges)

    def number_of_nodes(self):
        return len(self._nodes)

    def number_of_edges(self):
        return len(self._edges)


class FrozenTopology:
    def __init__(self, name, nodes, node_indices, k_paths, edges, modulations, pair_paths):
        """``edges``: list of (a, b, index, id, length); ``pair_paths``: {(a, b): [(path_id, hops, length,
        se, node_list), ...]} for node-order pairs a < b."""
        self.name = name
        self.nodes: List[str] = list(nodes)
        self.node_indices: List[str] = list(node_indices)
        if self.nodes != self.node_indices:
            # optical_network_env.py:197-207 samples by nodes() position and indexes by node_indices
            raise ValueError("topology.nodes() order differs from graph['node_indices']; unsupported")
        self.k_paths = int(k_paths)
        self.edges = [(str(a), str(b), int(i), int(d), float(l)) for a, b, i, d, l in edges]
        self.modulations = [Modulation(n, ml, int(se)) for n, ml, se in modulations]
        N, E = len(self.nodes), len(self.edges)
        if N > MAX_NODES or E > MAX_LINKS:
            raise ValueError(f"topology too large for the device tables (N={N}, E={E})")
        if sorted(e[2] for e in self.edges) != list(range(E)):
            raise ValueError("edge 'index' attributes must be 0..E-1")
        for a, b, idx, eid, _ in self.edges:
            if idx != eid:
                # rmsa_env.py:750 indexes by "id", everything else by "index" 
                raise ValueError("edge 'id' != 'index'; the reference's get_available_slots breaks here")
        self.num_nodes, self.num_links = N, E
        node_pos = {n: i for i, n in enumerate(self.nodes)}
        link_index = {}
        for a, b, idx, _, _ in self.edges:
            link_index[a, b] = idx
            link_index[b, a] = idx
        mod_by_se = {m.spectral_efficiency: m for m in self.modulations}

        self.ksp: Dict[Tuple[str, str], List[Path]] = {}
        pair_base = np.full(N * N, -1, np.int32)
        pair_count = np.zeros(N * N, np.int32)
        hops, ses, lengths, link_off, links = [], [], [], [0], []
        gid = 0
        for (a, b), plist in pair_paths.items():
            objs = []
            ia, ib = node_pos[a], node_pos[b]
            pair_base[ia * N + ib] = pair_base[ib * N + ia] = gid
            pair_count[ia * N + ib] = pair_count[ib * N + ia] = len(plist)
            if len(plist) != self.k_paths:
                raise ValueError(f"pair {a}-{b} has {len(plist)} paths, expected k={self.k_paths}")
            for idp, (path_id, h, length, se, node_list) in enumerate(plist):
                node_list = tuple(str(n) for n in node_list)
                if h != len(node_list) - 1 or h > MAX_HOPS or h < 1:
                    raise ValueError(f"bad hop count {h} on path {path_id}")
                p = Path(path_id=int(path_id), node_list=node_list, hops=int(h), length=length,
                         best_modulation=mod_by_se.get(int(se), Modulation(f"SE{se}", length, int(se))), gid=gid)
                objs.append(p)
                hops.append(int(h)); ses.append(int(se)); lengths.append(float(length))
                for i in range(h):
                    links.append(link_index[node_list[i], node_list[i + 1]])
                link_off.append(len(links))
                gid += 1
            # the reference shares ONE list object between (a,b) and (b,a) (create_topology.py:136-137)
            self.ksp[a, b] = objs
            self.ksp[b, a] = objs
        self.num_paths = gid
        self.pair_path_base = pair_base
        self.pair_path_count = pair_count
        self.path_hops = np.asarray(hops, np.int32)
        self.path_se = np.asarray(ses, np.int32)
        self.path_length = np.asarray(lengths, np.float64)
        self.path_link_off = np.asarray(link_off, np.int32)
        self.path_links = np.asarray(links, np.int32)
        self.link_length = np.zeros(E, np.float64)
        self.link_ends = np.zeros((E, 2), np.int32)
        for a, b, idx, _, l in self.edges:
            self.link_length[idx] = l
            self.link_ends[idx] = (node_pos[a], node_pos[b])
        self._pair_paths = pair_paths
        # node ids along every path (CSR), for the neighbour-link "cut" metric of phy_rmsa_env.py:1123-1193
        node_off, node_ids = [0], []
        for (a, b), plist in pair_paths.items():
            for (_, _, _, _, node_list) in plist:
                node_ids.extend(node_pos[str(n)] for n in node_list)
                node_off.append(len(node_ids))
        self.path_node_off = np.asarray(node_off, np.int32)
        self.path_nodes = np.asarray(node_ids, np.int32)

    def cut_adjacency(self):
        """Per path: the links adjacent to the path's nodes that are not path links, with weight 1 at the two end
        nodes and 2 at interior nodes -- ``calculate_r_cut(..., modified=True)`` (``phy_rmsa_env.py:1140-1193``)
        reduces to  sum_j w_j * (1 - 2 * available[link_j, channel])  for a channel free on the path.
        Returns CSR arrays (offsets [num_paths+1], links, weights)."""
        N = self.num_nodes
        link_of = {}
        nbrs = [[] for _ in range(N)]
        for l, (a, b) in enumerate(self.link_ends):
            link_of[int(a), int(b)] = l
            link_of[int(b), int(a)] = l
            nbrs[int(a)].append(int(b))
            nbrs[int(b)].append(int(a))
        off, links, weights = [0], [], []
        for g in range(self.num_paths):
            nodes = [int(x) for x in self.path_nodes[self.path_node_off[g]:self.path_node_off[g + 1]]]
            on_path = set(nodes)
            for i, n in enumerate(nodes):
                w = 1 if i in (0, len(nodes) - 1) else 2
                for nk in nbrs[n]:
                    if nk not in on_path:
                        links.append(link_of[n, nk])
                        weights.append(w)
            off.append(len(links))
        return np.asarray(off, np.int32), np.asarray(links, np.int32), np.asarray(weights, np.int32)

    def cut_node_tables(self):
        """The cut metric through per-node free degrees (networks of at most 16 nodes).

        ``calculate_r_cut(modified=True)`` sums, over the links that join a path node to a node OFF the path, weight x
        (1 - 2 x available) with weight 1 at the path's end nodes and 2 inside (:meth:`cut_adjacency`).  With
        ``D[v, ch]`` = number of links at node ``v`` that are free on channel ``ch`` this is a dot product:

            sum_j w_j * available_j  =  c . D[:, ch]  -  (the path's own links, if free)  -  (chords, if free)

        where ``c[v]`` = 1 / 2 / 0 (end / interior / off-path node) and a chord is a link between two path nodes that is not
        a path link (weight ``c[u] + c[v]``).  When a channel is taken or returned on a whole path, ``D[:, ch]`` changes by
        exactly ``c`` (a node loses one free link per path link it touches).  Returns ``(records [num_paths, 32] uint8,
        degree [16] uint8)`` or ``None`` when the network does not fit (more than 16 nodes, a path with more than 5 chords).
        Record: bytes 0..15 ``c``, 16..17 sum of the adjacency weights (int16), 18..19 ``c . (path links per node)``
        (int16), 20 number of chords, 21..25 chord links, 26..30 chord weights."""
        N = self.num_nodes
        if N > 16 or self.num_links > 255:
            return None
        link_of, nbrs = {}, [[] for _ in range(N)]
        deg = np.zeros(16, np.uint8)
        for l, (a, b) in enumerate(self.link_ends):
            link_of[int(a), int(b)] = l
            link_of[int(b), int(a)] = l
            nbrs[int(a)].append(int(b))
            nbrs[int(b)].append(int(a))
            deg[int(a)] += 1
            deg[int(b)] += 1
        rec = np.zeros((self.num_paths, 32), np.uint8)
        for g in range(self.num_paths):
            nodes = [int(x) for x in self.path_nodes[self.path_node_off[g]:self.path_node_off[g + 1]]]
            on_path = set(nodes)
            c = {n: (1 if i in (0, len(nodes) - 1) else 2) for i, n in enumerate(nodes)}
            path_links = {link_of[nodes[i], nodes[i + 1]] for i in range(len(nodes) - 1)}
            wsum, chords = 0, {}
            for n in nodes:
                for nk in nbrs[n]:
                    if nk not in on_path:
                        wsum += c[n]
                    else:
                        l = link_of[n, nk]
                        if l not in path_links:
                            chords[l] = c[n] + c[nk]
            if len(chords) > 5:
                return None
            for n in nodes:
                rec[g, n] = c[n]
            cq = sum(c[nodes[i]] + c[nodes[i + 1]] for i in range(len(nodes) - 1))
            rec[g, 16:18] = np.frombuffer(np.int16(wsum).tobytes(), np.uint8)
            rec[g, 18:20] = np.frombuffer(np.int16(cq).tobytes(), np.uint8)
            rec[g, 20] = len(chords)
            for q, (l, w) in enumerate(sorted(chords.items())):
                rec[g, 21 + q] = l
                rec[g, 26 + q] = w
        return rec, deg

    def pair_table_rows(self, pairs):
        """[N*N] row of the QoT tables for every ordered node pair: first row whose (source, destination)
        node numbers match in either order (``phy_rmsa_env.py:562-565``); -1 where none does."""
        N = self.num_nodes
        out = np.full(N * N, -1, np.int32)
        pairs = np.asarray(pairs)
        for i, a in enumerate(self.nodes):
            for j, b in enumerate(self.nodes):
                if i == j:
                    continue
                ia, ib = int(a), int(b)
                m = np.where(((pairs[:, 0] == ia) & (pairs[:, 1] == ib)) | ((pairs[:, 0] == ib) & (pairs[:, 1] == ia)))[0]
                if len(m):
                    out[i * N + j] = m[0]
        return out

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_json(cls, path_or_dict) -> "FrozenTopology":
        if isinstance(path_or_dict, dict):
            d = path_or_dict
        else:
            with open(path_or_dict) as f:
                d = json.load(f)
        pair_paths = {}
        for key, plist in d["paths"].items():
            a, b = key.split(",")
            pair_paths[a, b] = [tuple(p) for p in plist]
        return cls(d["name"], d["nodes"], d["node_indices"], d["k_paths"], d["edges"], d["modulations"], pair_paths)

    @classmethod
    def from_graph(cls, g) -> "FrozenTopology":
        """Freeze a reference-style graph (``optical_network_env.py:30-31,59-62``; ``rmsa_env.py:71``)."""
        if isinstance(g, FrozenTopology):
            return g
        assert "ksp" in g.graph and "k_paths" in g.graph and "modulations" in g.graph
        nodes = [str(n) for n in g.nodes()]
        edges = [(str(a), str(b), int(g[a][b]["index"]), int(g[a][b]["id"]), float(g[a][b]["length"]))
                 for a, b in g.edges()]
        pair_paths = {}
        for i, a in enumerate(nodes):
            for j, b in enumerate(nodes):
                if i < j:
                    pair_paths[a, b] = [
                        (int(p.path_id), int(p.hops), float(p.length), int(p.best_modulation.spectral_efficiency),
                         [str(n) for n in p.node_list])
                        for p in g.graph["ksp"][a, b]
                    ]
        mods = [(m.name, m.maximum_length, m.spectral_efficiency) for m in g.graph["modulations"]]
        return cls(g.graph["name"], nodes, [str(n) for n in g.graph["node_indices"]], g.graph["k_paths"],
                   edges, mods, pair_paths)

    def to_json(self) -> dict:
        return {
            "name": self.name, "nodes": self.nodes, "node_indices": self.node_indices, "k_paths": self.k_paths,
            "edges": [list(e) for e in self.edges],
            "modulations": [[m.name, m.maximum_length, m.spectral_efficiency] for m in self.modulations],
            "paths": {f"{a},{b}": [list(p) for p in pl] for (a, b), pl in self._pair_paths.items()},
        }

    # ------------------------------------------------------------------ views
    def view(self) -> TopologyView:
        """A fresh graph-like object (the reference deep-copies its topology per env,
        ``optical_network_env.py:58``)."""
        graph = {"name": self.name, "ksp": self.ksp, "k_paths": self.k_paths,
                 "modulations": tuple(self.modulations), "node_indices": list(self.node_indices)}
        edges = [(a, b, {"index": idx, "id": eid, "length": l, "weight": 1}) for a, b, idx, eid, l in self.edges]
        return TopologyView(self.nodes, edges, graph)

    def packed_path_records(self) -> np.ndarray:
        """[num_paths, 16] uint8 device records: hops, se, link[0..13]."""
        rec = np.zeros((self.num_paths, 16), np.uint8)
        rec[:, 0] = self.path_hops
        rec[:, 1] = self.path_se
        for g in range(self.num_paths):
            lo, hi = self.path_link_off[g], self.path_link_off[g + 1]
            rec[g, 2:2 + hi - lo] = self.path_links[lo:hi]
        return rec


def selection_tables(node_request_probabilities, bit_rate_probabilities, num_nodes, bit_rates):
    """Cumulative-weight tables exactly as CPython's ``random.choices`` builds them from the
    reference's weights (``optical_network_env.py:197-206``, ``rmsa_env.py:104-114``):
    ``list(itertools.accumulate(weights))`` on float64, the destination weights being the source
    weights with the source zeroed and re-normalised by ``np.sum``."""
    from itertools import accumulate

    if node_request_probabilities is None:
        probs = np.full((num_nodes,), fill_value=1.0 / num_nodes)
    else:
        probs = np.asarray(node_request_probabilities, dtype=np.float64)
    assert len(probs) == num_nodes
    src_cum = np.array(list(accumulate(probs)), np.float64)
    dst_cum = np.zeros((num_nodes, num_nodes), np.float64)
    for s in range(num_nodes):
        p = np.copy(probs)
        p[s] = 0.0
        p = p / np.sum(p)
        dst_cum[s] = list(accumulate(p))
    if bit_rate_probabilities is None:
        bit_rate_probabilities = [1.0 / len(bit_rates) for _ in range(len(bit_rates))]
    assert len(bit_rate_probabilities) == len(bit_rates)
    br_cum = np.array(list(accumulate(bit_rate_probabilities)), np.float64)
    return probs, src_cum, dst_cum, br_cum
