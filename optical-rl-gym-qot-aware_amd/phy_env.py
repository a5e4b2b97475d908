"""Single-environment view of the QoT-aware environment with the reference's object surface.

``PhyRMSAEnv`` is a drop-in for ``optical_rl_gym.envs.phy_rmsa_env.PhyRMSAEnv`` (physical and virtual layer,
periodic defragmentation): same constructor kwargs, ``step((path, channels))`` returning the reference's 5-tuple
``(obs, reward, done, False, info)`` with the same info keys (``phy_rmsa_env.py:319-348``), and the attributes / query
methods its heuristic callbacks touch (``phy_rmsa_env.py:1254-1737``): ``is_channel_free``, ``calculate_r_cut``,
``calculate_r_spatial``, ``modulation_level``, ``connections_detail``, ``k_shortest_paths``, ``channel_state``,
``grooming``, ``topology[a][b]["index"]``, ``topology.graph["num_channel_resources"]``, ``current_service``.
The environment lives on the GPU (a :class:`BatchedPhyRMSAEnv` of batch 1); queries read device state through the C
ABI.  The heuristics at the bottom are this package's statements of the reference's callbacks on that surface.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import _lib
from .envs import RMSAEnv as _RMSAView
from .phy import PHY_DEFAULT_BIT_RATES, BatchedPhyRMSAEnv, encode_channels
from .topology import Path, Service


class _PhyGraph(dict):
    def __init__(self, static, env):
        super().__init__(static)
        self._env = env

    def __getitem__(self, key):
        if key == "available_channels":
            return self._env._available()
        return super().__getitem__(key)


class _ChannelState:
    """``env.channel_state[src_id, dst_id, k-path]`` (``phy_rmsa_env.py:117-125``): the list of (channel, used, free,
    capacity) tuples of partially used channels, read from the device on first use after every step."""

    def __init__(self, env):
        self._env = env

    def __getitem__(self, key):
        s, d, k = (int(x) for x in key)
        return list(self._env._channel_lists().get((s, d, k), []))


class PhyRMSAEnv:
    metadata = {"metrics": ["service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate",
                            "episode_bit_rate_blocking_rate"]}

    def __init__(self, topology=None, episode_length: int = 1000, load: float = 10,
                 mean_service_holding_time: float = 10800.0, num_spectrum_resources: int = 100,
                 bit_rate_selection: str = "discrete", bit_rates=PHY_DEFAULT_BIT_RATES, bit_rate_probabilities=None,
                 node_request_probabilities=None, seed: Optional[int] = None, allow_rejection: bool = False,
                 reset: bool = True, channel_width: float = 12.5, number_spectrum_channels: int = 80,
                 number_spectrum_channels_s_band: int = 108, l_band: bool = True, s_band: bool = True,
                 modulation_level=None, connections_detail=None, gsnr=None, defrag_period=None, number_moves=None,
                 metric: str = "cut", grooming: bool = True, device: int = 0, **_ignored):
        self._batched = BatchedPhyRMSAEnv(
            topology, 1, modulation_level=modulation_level, connections_detail=connections_detail, gsnr=gsnr,
            episode_length=episode_length, load=load, mean_service_holding_time=mean_service_holding_time,
            bit_rates=bit_rates, bit_rate_probabilities=bit_rate_probabilities,
            node_request_probabilities=node_request_probabilities, seed=seed, allow_rejection=allow_rejection,
            number_spectrum_channels=number_spectrum_channels,
            number_spectrum_channels_s_band=number_spectrum_channels_s_band, l_band=l_band, s_band=s_band,
            defrag_period=defrag_period, number_moves=number_moves, metric=metric, grooming=grooming, device=device)
        b = self._batched
        ft = b.topology
        self._ft = ft
        view = ft.view()
        view.graph = _PhyGraph(dict(view.graph, num_spectrum_resources=num_spectrum_resources,
                                    num_channel_resources=b.num_channels), self)
        self.topology = view
        self.k_paths = ft.k_paths
        self.k_shortest_paths = ft.ksp
        self.modulation_level = np.asarray(modulation_level)
        self.connections_detail = np.asarray(connections_detail)
        self.gsnr = np.asarray(gsnr)
        self.grooming = grooming
        self.metric = metric
        self.episode_length = b.episode_length
        self.num_spectrum_channels = number_spectrum_channels
        self.number_spectrum_channels_s_band = number_spectrum_channels_s_band
        self.allow_rejection = allow_rejection
        self.bit_rates = list(b.bit_rates)
        self.channel_state = _ChannelState(self)
        self.services_accepted_virtual = 0
        # BVT counters per band and node pair (phy_rmsa_env.py:153-156: 1 = C band, 0 = L band, 2 = S band), kept on the host
        self.bvts = np.zeros((3, ft.num_nodes, ft.num_nodes), dtype=int)
        self.current_service: Optional[Service] = None
        self._sync()

    # ------------------------------------------------------------------ plumbing
    def _sync(self):
        b = self._batched
        r = b.requests()[0]
        nodes = self._ft.nodes
        self.current_service = Service(int(r["service_id"]), nodes[r["src"]], int(r["src"]), destination=nodes[r["dst"]],
                                       destination_id=int(r["dst"]), arrival_time=float(r["arrival_time"]),
                                       holding_time=float(r["holding_time"]), bit_rate=int(r["bit_rate"]))
        for name, arr in b.counters().items():
            setattr(self, name, int(arr[0]))
        self.current_time = float(b.current_time()[0])
        self._avail = None
        self._cs = None

    def _channel_lists(self):
        if self._cs is None:
            self._cs = self._batched.channel_state(0)
        return self._cs

    def _available(self):
        if self._avail is None:
            self._avail = self._batched.available_channels()[0].astype(np.int64)
        return self._avail

    def _links(self, path: Path):
        return [self.topology[path.node_list[i]][path.node_list[i + 1]]["index"] for i in range(len(path.node_list) - 1)]

    # ------------------------------------------------------------------ reference query surface
    def is_channel_free(self, path: Path, channel_number: int) -> bool:
        """``phy_rmsa_env.py:1029-1035``"""
        return bool(self._available()[self._links(path), channel_number].all())

    def is_path_free_on_channels(self, path: Path, selected_channels) -> bool:
        """``phy_rmsa_env.py:1019-1027``"""
        av = self._available()
        return all(bool(av[self._links(path), c[0]].all()) for c in selected_channels)

    rle = staticmethod(_RMSAView.rle)

    def calculate_r_cut(self, channel_number, link_indexes, defrag_flag: bool = False, path: Path = None, modified=False):
        """``phy_rmsa_env.py:1123-1193``"""
        col = self._available()[:, channel_number]
        after = col.copy()
        after[list(link_indexes)] = 1 if defrag_flag else 0
        if not modified:
            return int(np.sum(self.rle(col)[1]) - np.sum(self.rle(after)[1]))
        nodes = list(path.node_list)

        def cuts(t):
            n = 0
            for i, node in enumerate(nodes):
                for nk in self.topology[node]:
                    if nk in nodes:
                        continue
                    a = t[self.topology[node][nk]["index"]]
                    if i == len(nodes) - 1:
                        n += abs(t[self.topology[nodes[i - 1]][node]["index"]] - a)
                    elif i == 0:
                        n += abs(t[self.topology[node][nodes[1]]["index"]] - a)
                    else:
                        n += abs(t[self.topology[node][nodes[i + 1]]["index"]] - a) + \
                            abs(t[self.topology[nodes[i - 1]][node]["index"]] - a)
            return n
        return int(cuts(col) - cuts(after))

    def calculate_r_spatial(self, channel_number, link_indexes, defrag_flag: bool = False):
        """``phy_rmsa_env.py:1085-1108``"""
        def rss(t):
            _, values, lengths = self.rle(t)
            free = lengths[values == 1]
            return np.sqrt(np.sum(free ** 2)) / (np.sum(free) + 1)
        col = self._available()[:, channel_number]
        after = col.copy()
        after[list(link_indexes)] = 1 if defrag_flag else 0
        return rss(after) - rss(col)

    # ------------------------------------------------------------------ gym surface
    def observation(self):
        return {"topology": self.topology, "current_service": self.current_service}

    def reward(self):
        return 1 if self.current_service.accepted else 0

    def render(self, mode="human"):
        return

    def seed(self, seed=None):
        """``optical_network_env.py:266-271``; see ``BatchedRMSAEnv.reseed`` for why this is refused."""
        raise NotImplementedError(
            "seed() after construction is not reproduced: the reference keeps drawing the BIT RATE from the generator object of "
            "construction time (functools.partial(self.rng.choices, ...)) while the other four draws of a request come from "
            "Random(seed) -- two generators per environment.  Pass seed= to the constructor, or call reseed() on the batched "
            "environment for a fresh generator for all draws (not the reference's stream).")

    def reset(self, only_episode_counters: bool = True):
        self._batched.reset(only_episode_counters)
        self._sync()
        return self.observation()

    def step(self, action):
        """``phy_rmsa_env.py:272-424`` -> (observation, reward, done, False, info)"""
        path, channels = action[0], action[1]
        served = self.current_service
        ap = np.array([path], np.int32)
        ac = np.full((1, _lib.PHY_MAX_CHANNELS), -1, np.int16)
        encode_channels(channels, ac[0])
        if path > 10:
            self.services_accepted_virtual += 1
        r = self._batched.run("external", 1, act_path=ap, act_channels=ac,
                              outputs=("accepted", "done", "number_cuts_total", "rss_total_metric", "defrag_counters"))
        served.accepted = bool(r["accepted"][0, 0])
        if served.accepted and path <= 10:
            # _provision_path (phy_rmsa_env.py:603-608): one transceiver per channel lit, by the band of its index -- with the
            # reference's bounds (index == number_spectrum_channels counts as C band, 2 * number_spectrum_channels as L band)
            n, ns = self.num_spectrum_channels, self.number_spectrum_channels_s_band
            for ch in channels:
                c0 = int(ch[0])
                band = 1 if c0 <= n else 0 if c0 <= 2 * n else 2 if c0 < 2 * n + ns else None
                if band is not None:
                    self.bvts[band][served.source_id][served.destination_id] += 1
        b = self._batched
        c = {k: int(v[0]) for k, v in b.counters().items()}
        nxt = int(b.requests()[0]["bit_rate"])
        c["services_processed"] -= 1          # info is built before _next_service (phy_rmsa_env.py:319-351)
        c["episode_services_processed"] -= 1
        c["bit_rate_requested"] -= nxt
        c["episode_bit_rate_requested"] -= nxt
        st = b.episode_stats()[0]
        phys, chans = int(st["physical_services_accepted"]), int(st["channels_accepted"])
        info = {
            "service_blocking_rate": (c["services_processed"] - c["services_accepted"]) / c["services_processed"],
            "episode_service_blocking_rate": (c["episode_services_processed"] - c["episode_services_accepted"])
            / c["episode_services_processed"],
            "bit_rate_blocking_rate": (c["bit_rate_requested"] - c["bit_rate_provisioned"]) / c["bit_rate_requested"],
            "episode_bit_rate_blocking_rate": (c["episode_bit_rate_requested"] - c["episode_bit_rate_provisioned"])
            / c["episode_bit_rate_requested"],
            "number_cuts_total": float(r["number_cuts_total"][0, 0]),
            "rss_total_metric": float(r["rss_total_metric"][0, 0]),
            "total_path_length": float(st["total_path_length"]) / (phys + 1),
            "num_moves": int(r["defrag_counters"][0, 0, 0]) / 2 + int(r["defrag_counters"][0, 0, 1]),
            "num_moves_groom": int(r["defrag_counters"][0, 0, 1]), "num_defrag_cycle": int(r["defrag_counters"][0, 0, 2]),
            "avrage_gsnr": float(st["total_gsnr"]) / (chans + 1),
            # the reference's accumulator wraps at 256 under NumPy >= 2 (SURVEY 8c caveat 2); this is the true mean
            "average_mod_level": int(st["total_modulation_level"]) / (chans + 1),
            "average_path_index": int(st["total_path_index"]) / (phys + 1),
            "path_index": int(st["total_path_index"]),
            "physical_paths": phys,
        }
        done = bool(r["done"][0, 0])
        self._sync()
        return self.observation(), (1 if served.accepted else 0), done, False, info

    def close(self):
        self._batched.close()


def _table_id(env) -> int:
    cd = env.connections_detail
    s, d = int(env.current_service.source), int(env.current_service.destination)
    a, b = cd[:, 0].astype(int), cd[:, 1].astype(int)
    return int(np.where(((a == s) & (b == d)) | ((a == d) & (b == s)))[0][0])


def use_existing_channels(env) -> Tuple[int, list]:
    """``phy_rmsa_env.py:1650-1673``: serve the request from residual capacity of channels already lit between the same
    (source, destination) on k-path idp; ``(-3, [])`` when no path's list covers the bit rate.  Like the reference the
    running remainder is NOT restarted when a path's list turns out not to cover it."""
    sv = env.current_service
    unassigned, selected = sv.bit_rate, []
    for idp in range(len(env.k_shortest_paths[sv.source, sv.destination])):
        state = env.channel_state[sv.source_id, sv.destination_id, idp]
        if sum(c[2] for c in state) >= unassigned / 100:
            for ch, _used, free, cap in state:
                if free > 0:
                    unassigned -= free * 100
                    if unassigned <= 0:
                        selected.append((ch, free + unassigned / 100, unassigned / -100, cap, True))
                        return (idp, selected)
                    selected.append((ch, free, 0, cap, True))
    return (-3, [])


def _free_channel_rows(env, metric=None):
    """Per candidate path the free channels as (level, metric, channel, idp) in channel order."""
    table_id = _table_id(env)
    rows = []
    for idp, path in enumerate(env.k_shortest_paths[env.current_service.source, env.current_service.destination]):
        links = [env.topology[path.node_list[i]][path.node_list[i + 1]]["index"] for i in range(len(path.node_list) - 1)]
        row = []
        for ch in range(env.topology.graph["num_channel_resources"]):
            if env.is_channel_free(path, ch):
                level = int(env.modulation_level[table_id][ch][idp])
                m = 0 if metric is None else metric(env, ch, links, path)
                row.append((level, m, ch, idp))
        rows.append(row)
    return rows


def _take_channels(env, rows, pick):
    """Greedy cover of the bit rate from the row ``pick`` selects; a row that cannot cover it is dropped."""
    rows = [r for r in rows]
    while True:
        rows = [r for r in rows if r]
        best = pick(rows)
        if best is None:
            return (-2, [])
        unassigned, selected = env.current_service.bit_rate, []
        for level, _, ch, idp in rows[best]:
            unassigned -= level * 100
            if unassigned <= 0:
                selected.append((ch, level + unassigned / 100, unassigned / -100, level, False))
                return (idp, selected)
            selected.append((ch, level, 0, level, False))
        rows.pop(best)


def _with_virtual_layer(env, always):
    if always or env.grooming:
        idp, chans = use_existing_channels(env)
        if idp != -3:
            return (idp + 20, chans)
    return None


def sapff_rmsa(env) -> Tuple[int, list]:
    """``phy_rmsa_env.py:1676-1737``: virtual layer first, then the first path with a free channel, channels in index order."""
    v = _with_virtual_layer(env, True)
    if v:
        return v
    return _take_channels(env, _free_channel_rows(env), lambda rows: 0 if rows else None)


def phy_aware_sapbm_rmsa(env) -> Tuple[int, list]:
    """``phy_rmsa_env.py:1254-1314``: virtual layer first, then the first path with a free channel, best modulation first."""
    v = _with_virtual_layer(env, True)
    if v:
        return v
    rows = [sorted(r, key=lambda x: (-x[0], x[2])) for r in _free_channel_rows(env)]
    return _take_channels(env, rows, lambda rows: 0 if rows else None)


def phy_aware_bmff_rmsa(env) -> Tuple[int, list]:
    """``phy_rmsa_env.py:1317-1372``: virtual layer first, then the path whose best free channel has the highest
    modulation level (ties: the earlier path), channels by (level desc, index)."""
    v = _with_virtual_layer(env, True)
    if v:
        return v
    rows = [sorted(r, key=lambda x: (-x[0], x[2])) for r in _free_channel_rows(env)]

    def pick(rows):
        best, head = None, float("-inf")
        for i, row in enumerate(rows):
            if row[0][0] > head:
                best, head = i, row[0][0]
        return best
    return _take_channels(env, rows, pick)


def _pick_level_metric(rows):
    best, head = None, (float("-inf"), float("-inf"))
    for i, row in enumerate(rows):
        if (row[0][0], row[0][1]) > head:
            best, head = i, (row[0][0], row[0][1])
    return best


def _cut_metric(e, ch, links, path):
    return e.calculate_r_cut(ch, links, False, path, True)


def _rss_metric(e, ch, links, path):
    return e.calculate_r_spatial(ch, links, False)


def _faff(env, metric):
    v = _with_virtual_layer(env, True)
    if v:
        return v
    rows = [sorted(r, key=lambda x: -x[1]) for r in _free_channel_rows(env, metric)]

    def pick(rows):
        best, head = None, float("-inf")
        for i, row in enumerate(rows):
            if row[0][1] > head:
                best, head = i, row[0][1]
        return best
    return _take_channels(env, rows, pick)


def phy_aware_faff_rmsa(env) -> Tuple[int, list]:
    """``phy_rmsa_env.py:1508-1569``: virtual layer first, then fragmentation-aware first: channels by cut metric
    (desc, ties in index order) whatever their modulation level, the path whose best channel has the highest metric."""
    return _faff(env, _cut_metric)


def phy_aware_faff_rss_rmsa(env) -> Tuple[int, list]:
    """``phy_rmsa_env.py:1572-1647``: as faff with the RSS metric."""
    return _faff(env, _rss_metric)


def phy_aware_bmfa_rss_rmsa(env) -> Tuple[int, list]:
    """``phy_rmsa_env.py:1441-1505``: as bmfa with the RSS metric (``calculate_r_spatial``)."""
    v = _with_virtual_layer(env, False)
    if v:
        return v
    rows = _free_channel_rows(env, _rss_metric)
    return _take_channels(env, [sorted(r, key=lambda x: (-x[0], -x[1])) for r in rows], _pick_level_metric)


def phy_aware_bmfa_rmsa(env) -> Tuple[int, list]:
    """Best-modulation, fragmentation-aware (cut metric) channel selection, ``phy_rmsa_env.py:1375-1438`` with
    the virtual layer first when ``env.grooming``: per path the free channels sorted by (level desc, cut metric desc);
    the row with the best head wins; channels are taken in order until the bit rate is covered (the last one partially)."""
    v = _with_virtual_layer(env, False)
    if v:
        return v
    table_id = _table_id(env)
    rows = []
    for idp, path in enumerate(env.k_shortest_paths[env.current_service.source, env.current_service.destination]):
        links = [env.topology[path.node_list[i]][path.node_list[i + 1]]["index"] for i in range(len(path.node_list) - 1)]
        row = []
        for ch in range(env.topology.graph["num_channel_resources"]):
            if env.is_channel_free(path, ch):
                level = int(env.modulation_level[table_id][ch][idp])
                row.append((level, env.calculate_r_cut(ch, links, False, path, True), ch, idp))
        rows.append(sorted(row, key=lambda x: (-x[0], -x[1])))
    while True:
        best, head = None, (float("-inf"), float("-inf"))
        for i, row in enumerate(rows):
            if row and (row[0][0], row[0][1]) > head:
                best, head = i, (row[0][0], row[0][1])
        if best is None:
            return (-2, [])
        unassigned, selected = env.current_service.bit_rate, []
        for level, _, ch, idp in rows[best]:
            unassigned -= level * 100
            if unassigned <= 0:
                selected.append((ch, level + unassigned / 100, unassigned / -100, level, False))
                return (idp, selected)
            selected.append((ch, level, 0, level, False))
        rows.pop(best)
