"""BatchedPhyRMSAEnv: B independent QoT-aware environments (``PhyRMSAEnv``, ``phy_rmsa_env.py:20``) on one
MI355X: physical layer and virtual ("grooming") layer.  Constructor kwargs are the reference's
(``phy_rmsa_env.py:30-58``); ``modulation_level`` / ``gsnr`` are the ``(pairs, channels, k)`` tables and
``connections_detail`` the table's (source, destination) node numbers per row (an ``[rows, 2]`` int array, or the
reference's MATLAB object array).

Device policies (``run(policy, ...)``): ``bmfa`` / ``bmfa_rss`` (``phy_rmsa_env.py:1375,1441``; they consult the
virtual layer only when ``grooming=True``), ``sapff`` / ``bmff`` / ``sapbm`` / ``faff`` / ``faff_rss``
(``:1676,1317,1254,1508,1572``; they always try ``use_existing_channels`` first, like the reference) and ``external``.  ``defrag_period`` / ``number_moves`` / ``metric``
switch on the periodic defragmentation (``phy_rmsa_env.py:355-417``), run inside the step kernel.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .batched import COUNTER_NAMES, REQUEST_DTYPE, _check_buffer, _ptr
from .topology import FrozenTopology, selection_tables

PHY_DEFAULT_BIT_RATES = (100, 200, 300, 400, 500, 600)  # phy_rmsa_env.py:38

EPISODE_STATS_DTYPE = np.dtype([("total_path_length", np.float64), ("total_gsnr", np.float64),
                                ("total_path_index", np.int64), ("total_modulation_level", np.int64),
                                ("channels_accepted", np.int64), ("physical_services_accepted", np.int64),
                                ("episodes_done", np.int64), ("queue_overflow", np.int64),
                                ("counted_moves", np.int64), ("counted_moves_groom", np.int64),
                                ("counted_defrag_cycles", np.int64)])


def _pairs_from_connections_detail(cd):
    cd = np.asarray(cd)
    if cd.dtype == object:  # scipy.io.loadmat cell array: column 0 / 1 hold 1x1 arrays
        return np.array([[int(np.asarray(r[0]).ravel()[0]), int(np.asarray(r[1]).ravel()[0])] for r in cd], np.int32)
    return np.ascontiguousarray(cd[:, :2], dtype=np.int32)


def encode_channels(selected_channels, out_row):
    """The reference's ``selected_channels`` tuples ``(channel, used, free, capacity, virtual)`` (``phy_rmsa_env.py:
    1364-1366``) -> one ``act_channels`` row: channel | used << 9 (``used`` in 100 Gb/s units; bare ints = whole channel)."""
    out_row[:] = -1
    for q, c in enumerate(selected_channels):
        if isinstance(c, (tuple, list, np.ndarray)):
            out_row[q] = int(c[0]) | (int(round(float(c[1]))) << 9 if len(c) > 1 else 0)
        else:
            out_row[q] = int(c)
    return out_row


class BatchedPhyRMSAEnv:
    def __init__(self, topology, batch_size: int, *, modulation_level, connections_detail, gsnr,
                 episode_length: int = 1000, load: float = 10, mean_service_holding_time: float = 10800.0,
                 bit_rates: Sequence[int] = PHY_DEFAULT_BIT_RATES, bit_rate_probabilities=None,
                 node_request_probabilities=None, seed: Optional[int] = None, seeds=None,
                 allow_rejection: bool = False, number_spectrum_channels: int = 80,
                 number_spectrum_channels_s_band: int = 108, l_band: bool = True, s_band: bool = True,
                 defrag_period=None, number_moves=None, metric: str = "cut", grooming: bool = False,
                 queue_capacity: int = 0, channel_state_capacity: int = 0, defrag_capacity: int = 0, device: int = 0,
                 gn_gate=None, **_ignored):
        if defrag_period and number_moves is None:
            raise ValueError("defrag_period needs number_moves (the reference compares against it, phy_rmsa_env.py:358)")
        self.L = _lib.load()
        self.topology = FrozenTopology.from_graph(topology)
        t = self.topology
        self.batch_size = int(batch_size)
        self.episode_length = int(episode_length)
        self.k_paths = t.k_paths
        self.bit_rates = [int(b) for b in bit_rates]
        self.allow_rejection = bool(allow_rejection)
        # optical_network_env.py:78-102
        if s_band:
            self.num_channels = 2 * number_spectrum_channels + number_spectrum_channels_s_band
        elif l_band:
            self.num_channels = 2 * number_spectrum_channels
        else:
            self.num_channels = number_spectrum_channels
        self.load, self.mean_service_holding_time = load, mean_service_holding_time
        self.mean_service_inter_arrival_time = 1 / float(load / float(mean_service_holding_time))
        self.node_request_probabilities, src_cum, dst_cum, br_cum = selection_tables(
            node_request_probabilities, bit_rate_probabilities, t.num_nodes, self.bit_rates)
        self.rand_seed = 41 if seed is None else int(seed)
        mod = np.ascontiguousarray(modulation_level, dtype=np.uint8)
        gs = np.ascontiguousarray(gsnr, dtype=np.float64)
        assert mod.shape == gs.shape and mod.shape[1] >= self.num_channels
        if mod.shape[1] != self.num_channels:
            mod, gs = np.ascontiguousarray(mod[:, :self.num_channels]), np.ascontiguousarray(gs[:, :self.num_channels])
        pairs = _pairs_from_connections_detail(connections_detail)
        adj_off, adj_link, adj_weight = t.cut_adjacency()

        self._keep = []

        def keep(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            self._keep.append(a)
            return a.ctypes.data_as(C.c_void_p)

        ct = _lib.Topology()
        ct.num_nodes, ct.num_links, ct.k_paths, ct.num_paths = t.num_nodes, t.num_links, t.k_paths, t.num_paths
        for name, dt in (("pair_path_base", np.int32), ("pair_path_count", np.int32), ("path_hops", np.int32),
                         ("path_se", np.int32), ("path_length", np.float64), ("path_link_off", np.int32),
                         ("path_links", np.int32)):
            setattr(ct, name, keep(getattr(t, name), dt))
        cc = _lib.PhyConfig()
        cc.num_channels, cc.episode_length, cc.num_bit_rates = self.num_channels, self.episode_length, len(self.bit_rates)
        cc.k_table, cc.num_table_rows, cc.queue_capacity = mod.shape[2], mod.shape[0], int(queue_capacity)
        cc.grooming, cc.channel_state_capacity = (1 if grooming else 0), int(channel_state_capacity)
        self.grooming = bool(grooming)
        cc.defrag_period, cc.number_moves = int(defrag_period or 0), int(number_moves or 0)
        cc.defrag_metric, cc.defrag_capacity = (0 if metric == "cut" else 1), int(defrag_capacity)
        self.defrag_period, self.number_moves, self.metric = defrag_period, number_moves, metric
        cc.arrival_lambda = 1 / self.mean_service_inter_arrival_time
        cc.holding_lambda = 1 / self.mean_service_holding_time
        cc.bit_rates = keep(self.bit_rates, np.int32)
        cc.bit_rate_cum = keep(br_cum, np.float64)
        cc.src_cum = keep(src_cum, np.float64)
        cc.dst_cum = keep(dst_cum, np.float64)
        cc.pair_table_row = keep(t.pair_table_rows(pairs), np.int32)
        cc.modulation_level = keep(mod, np.uint8)
        cc.gsnr = keep(gs, np.float64)
        cc.adj_off, cc.adj_link, cc.adj_weight = keep(adj_off, np.int32), keep(adj_link, np.int32), keep(adj_weight, np.int32)
        # the cut metric as byte dot products over per-node free degrees (networks of at most 16 nodes); ORLG_PHY_NODEVEC=0
        # keeps the adjacency-list evaluation (identical results: tests/test_gpu_phy.py runs both)
        import os
        nv = t.cut_node_tables() if os.environ.get("ORLG_PHY_NODEVEC", "1") != "0" else None
        if nv is not None:
            cc.path_node_weights, cc.node_degree = keep(nv[0], np.uint8), keep(nv[1], np.uint8)
            cc.link_ends = keep(np.asarray(t.link_ends, np.int32).reshape(-1, 2), np.int32)
        # GN-model admission check of the chosen channels (osnr.gn_gate_parameters; not in the reference: include/orlg.h)
        self.gn_gate = gn_gate
        if gn_gate is not None:
            gg = _lib.GnGate()
            gg.launch_power_w = float(gn_gate["launch_power_w"])
            gg.channel_bandwidth_hz = float(gn_gate["channel_bandwidth_hz"])
            gg.attenuation_normalized = float(gn_gate["attenuation_normalized"])
            gg.noise_figure = float(gn_gate["noise_figure"])
            cf = np.ascontiguousarray(gn_gate["channel_center_frequency_hz"], np.float64)
            ns, sl = np.ascontiguousarray(gn_gate["link_num_spans"], np.int32), np.ascontiguousarray(gn_gate["link_span_length_km"], np.float64)
            if cf.shape != (self.num_channels,) or ns.shape != (t.num_links,) or sl.shape != (t.num_links,):
                raise ValueError("gn_gate: channel_center_frequency_hz [num_channels], link_num_spans / link_span_length_km [num_links]")
            gg.channel_center_frequency_hz, gg.link_num_spans, gg.link_span_length_km = keep(cf, np.float64), keep(ns, np.int32), keep(sl, np.float64)
            thr = np.ascontiguousarray(gn_gate["thresholds_db"], np.float64)
            gg.thresholds_db, gg.num_thresholds = keep(thr, np.float64), len(thr)
            self._keep.append(gg)
            cc.gn_gate = C.cast(C.pointer(gg), C.c_void_p)
        seeds_ptr = None
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
            seeds_ptr = seeds.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        _lib.check(self.L.orlg_phy_create(C.byref(ct), C.byref(cc), self.batch_size, seeds_ptr,
                                          C.c_uint64(self.rand_seed), int(device), C.byref(h)))
        self.h = h
        self.words_per_link = self.L.orlg_phy_words_per_link(self.h)
        self.node_vectors = bool(self.L.orlg_phy_node_vectors(self.h))   # cut metric through node-degree vectors (include/orlg.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.orlg_phy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _lib.check(self.L.orlg_phy_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def synchronize(self):
        _lib.check(self.L.orlg_phy_synchronize(self.h))

    def reset(self, only_episode_counters: bool = True):
        _lib.check(self.L.orlg_phy_reset(self.h, 1 if only_episode_counters else 0))

    def last_kernel(self) -> str:
        """Name, template arguments and launch shape of the kernel behind the last ``run`` / ``reset``."""
        buf = C.create_string_buffer(128)
        _lib.check(self.L.orlg_phy_last_kernel(self.h, buf, 128))
        return buf.value.decode()

    def reseed(self, seed=None, seeds=None):
        """A fresh generator for every environment: see ``BatchedRMSAEnv.reseed`` (NOT the reference's ``seed()``)."""
        if seeds is not None:
            sa = np.ascontiguousarray(seeds, np.uint64)
            if sa.shape != (self.batch_size,):
                raise ValueError(f"seeds: shape {sa.shape}, expected ({self.batch_size},)")
            _lib.check(self.L.orlg_phy_reseed(self.h, _ptr(sa), 0))
        else:
            _lib.check(self.L.orlg_phy_reseed(self.h, None, int(41 if seed is None else seed)))

    def run(self, policy: str, n_steps: int = 1, *, act_path=None, act_channels=None, auto_reset: bool = False,
            outputs: Sequence[str] = (), out=None):
        """``n_steps`` x (policy -> PhyRMSAEnv.step).  ``policy='external'``: ``act_path`` [B] int32 (-2 = blocked,
        0..k-1 physical, 20 + k-path virtual layer) and ``act_channels`` [B, 14] int16 (-1 padded; entry = channel |
        used << 9, see :func:`encode_channels`).  Returns the requested per-step arrays [n_steps, B(, ...)]; ``out`` may
        supply preallocated numpy arrays or torch tensors by name (device tensors are written without staging)."""
        B = self.batch_size
        io = _lib.PhyStepIO()
        res = {}
        names = list(outputs) + [k for k in (out or {}) if k not in outputs]
        for name in names:
            if name not in _lib.PHY_STEP_IO_DTYPES:
                raise KeyError(f"unknown step output {name!r}")
            shape = {"request": (n_steps, B, 4), "channels": (n_steps, B, _lib.PHY_MAX_CHANNELS),
                     "channels_used": (n_steps, B, _lib.PHY_MAX_CHANNELS),
                     "defrag_counters": (n_steps, B, 3)}.get(name, (n_steps, B))
            if out is not None and name in out:
                res[name] = _check_buffer(f"out[{name!r}]", out[name], shape, _lib.PHY_STEP_IO_DTYPES[name])
            else:
                res[name] = np.zeros(shape, dtype=_lib.PHY_STEP_IO_DTYPES[name])
            setattr(io, name, _ptr(res[name]))
        ap = ac = None
        if policy == "external":
            if act_path is None or act_channels is None:
                raise ValueError("policy 'external' needs act_path and act_channels")
            # host arrays: integers only (as BatchedRMSAEnv.run), and every value must survive the cast to the ABI's types --
            # a float or an out-of-range entry is refused, not truncated or wrapped
            def _as(name, a, dt):
                a = np.asarray(a)
                if a.dtype.kind not in "iu":
                    raise TypeError(f"{name} must be an integer array, got {a.dtype}")
                info = np.iinfo(dt)
                if a.size and (a.min() < info.min or a.max() > info.max):
                    raise ValueError(f"{name} has values outside {np.dtype(dt).name}")
                return np.ascontiguousarray(a, dt)
            if not hasattr(act_path, "data_ptr"):
                act_path = _as("act_path", act_path, np.int32)
            if not hasattr(act_channels, "data_ptr"):
                act_channels = _as("act_channels", act_channels, np.int16)
            _check_buffer("act_path", act_path, (B,), np.int32)
            _check_buffer("act_channels", act_channels, (B, _lib.PHY_MAX_CHANNELS), np.int16)
            ap, ac = _ptr(act_path), _ptr(act_channels)
        _lib.check(self.L.orlg_phy_step(self.h, _lib.PHY_POLICIES[policy], int(n_steps), ap, ac,
                                        1 if auto_reset else 0, C.byref(io)))
        return res

    def requests(self):
        a = np.zeros(self.batch_size, REQUEST_DTYPE)
        _lib.check(self.L.orlg_phy_get_requests(self.h, _ptr(a)))
        return a

    def counters(self):
        a = np.zeros((self.batch_size, 8), np.int64)
        _lib.check(self.L.orlg_phy_get_counters(self.h, _ptr(a)))
        return {n: a[:, i].copy() for i, n in enumerate(COUNTER_NAMES)}

    def current_time(self):
        a = np.zeros(self.batch_size, np.float64)
        _lib.check(self.L.orlg_phy_get_current_time(self.h, _ptr(a)))
        return a

    def num_running(self):
        a = np.zeros(self.batch_size, np.int32)
        _lib.check(self.L.orlg_phy_get_num_running(self.h, _ptr(a)))
        return a

    def episode_stats(self):
        a = np.zeros(self.batch_size, EPISODE_STATS_DTYPE)
        _lib.check(self.L.orlg_phy_get_episode_stats(self.h, _ptr(a)))
        return a

    def info(self):
        """The per-episode ratios of the info dict (``phy_rmsa_env.py:339-347``) for every env."""
        s = self.episode_stats()
        phys, chans = s["physical_services_accepted"], s["channels_accepted"]
        return {"total_path_length": s["total_path_length"] / (phys + 1),
                "avrage_gsnr": s["total_gsnr"] / (chans + 1),
                "average_mod_level": s["total_modulation_level"] / (chans + 1),
                "average_path_index": s["total_path_index"] / (phys + 1),
                "path_index": s["total_path_index"], "physical_paths": phys}

    def available_channels(self):
        """topology.graph["available_channels"] for every env: [B, E, C] uint8 (1 = free)."""
        E, W = self.topology.num_links, self.words_per_link
        w = np.zeros((self.batch_size, E, W), np.uint64)
        _lib.check(self.L.orlg_phy_get_occupancy(self.h, _ptr(w)))
        bits = np.unpackbits(w.view(np.uint8), axis=-1, bitorder="little")
        return bits.reshape(self.batch_size, E, -1)[:, :, :self.num_channels]

    def channel_state(self, env_index: int = 0):
        """``env.channel_state`` of one env (``phy_rmsa_env.py:117-125``): dict (src_id, dst_id, k-path) -> list of
        (channel, used, free, capacity) tuples in list order (100 Gb/s units); empty lists are omitted."""
        t = self.topology
        lists = t.num_nodes * t.num_nodes * t.k_paths
        cap = self.L.orlg_phy_channel_state_capacity(self.h)
        ent = np.zeros((lists, cap), np.uint32)
        n = np.zeros(lists, np.uint8)
        rc = self.L.orlg_phy_get_channel_state(self.h, int(env_index), _ptr(ent), _ptr(n))
        if rc < 0:
            _lib.check(rc)
        out = {}
        for key in np.nonzero(n)[0]:
            e = ent[key, :n[key]].astype(np.int64)
            s, rem = divmod(int(key), t.num_nodes * t.k_paths)
            d, k = divmod(rem, t.k_paths)
            out[(s, d, k)] = [(int(x & 0x1ff), int((x >> 9) & 0x1f), int((x >> 14) & 0x1f), int((x >> 19) & 0x1f)) for x in e]
        return out

    def save_state(self):
        """Snapshot of the complete simulation state of the batch (a uint8 array): checkpoint / resume, env cloning."""
        n = self.L.orlg_phy_state_size(self.h)
        if n < 0:
            _lib.check(int(n))
        buf = np.empty(int(n), np.uint8)
        _lib.check(self.L.orlg_phy_save_state(self.h, _ptr(buf)))
        return buf

    def load_state(self, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        assert buf.size == self.L.orlg_phy_state_size(self.h), "snapshot of a differently configured batch"
        _lib.check(self.L.orlg_phy_load_state(self.h, _ptr(buf)))

    def reduce_counters(self):
        a = np.zeros(16, np.int64)
        _lib.check(self.L.orlg_phy_reduce_counters(self.h, _ptr(a)))
        d = {n: int(a[i]) for i, n in enumerate(COUNTER_NAMES)}
        d["episodes_done"], d["num_envs"] = int(a[8]), int(a[9])
        return d, a
