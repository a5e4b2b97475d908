"""ctypes loader for the CPU oracle (oracle/orlg_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
It takes plain numpy tables (the frozen topology CSR arrays and the cumulative-weight tables) so
that it does not depend on the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_BIT_RATES = 256

POLICY = {"sp_ff": 0, "sap_ff": 1, "llp_ff": 2, "deeprmsa_sp_ff": 3, "deeprmsa_sap_ff": 4, "external": -1,
          "deeprmsa_external": 5, "path_ff_external": 6}


class Topology(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("num_links", C.c_int32), ("k_paths", C.c_int32), ("num_paths", C.c_int32),
                ("pair_path_base", C.c_void_p), ("pair_path_count", C.c_void_p), ("path_hops", C.c_void_p),
                ("path_se", C.c_void_p), ("path_length", C.c_void_p), ("path_link_off", C.c_void_p),
                ("path_links", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("num_slots", C.c_int32), ("episode_length", C.c_int32), ("num_bit_rates", C.c_int32),
                ("j", C.c_int32), ("reward_mode", C.c_int32), ("pad0", C.c_int32),
                ("arrival_lambda", C.c_double), ("holding_lambda", C.c_double), ("channel_width", C.c_double),
                ("bit_rates", C.c_void_p), ("bit_rate_cum", C.c_void_p), ("src_cum", C.c_void_p),
                ("dst_cum", C.c_void_p)]


class Request(C.Structure):
    _fields_ = [("service_id", C.c_int32), ("src", C.c_int32), ("dst", C.c_int32), ("bit_rate", C.c_int32),
                ("arrival_time", C.c_double), ("holding_time", C.c_double)]


class StepResult(C.Structure):
    _fields_ = [("reward", C.c_double), ("done", C.c_int32), ("accepted", C.c_int32),
                ("service_blocking_rate", C.c_double), ("episode_service_blocking_rate", C.c_double),
                ("bit_rate_blocking_rate", C.c_double), ("episode_bit_rate_blocking_rate", C.c_double),
                ("network_compactness", C.c_double), ("network_compactness_difference", C.c_double),
                ("avg_link_compactness", C.c_double), ("avg_link_utilization", C.c_double),
                ("fairness", C.c_double), ("bit_rate_blocking", C.c_double * MAX_BIT_RATES)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "services_processed", "services_accepted", "episode_services_processed", "episode_services_accepted",
        "bit_rate_requested", "bit_rate_provisioned", "episode_bit_rate_requested", "episode_bit_rate_provisioned")]


TRACE_FIELDS = [
    ("service_id", np.int32), ("src", np.int32), ("dst", np.int32), ("bit_rate", np.int32),
    ("act_path", np.int32), ("act_slot", np.int32), ("arrival", np.float64), ("holding", np.float64),
    ("accepted", np.uint8), ("done", np.uint8), ("reward", np.float64),
    ("services_processed", np.int64), ("services_accepted", np.int64),
    ("episode_services_processed", np.int64), ("episode_services_accepted", np.int64),
    ("bit_rate_requested", np.int64), ("bit_rate_provisioned", np.int64),
    ("episode_bit_rate_requested", np.int64), ("episode_bit_rate_provisioned", np.int64),
    ("network_compactness", np.float64), ("network_compactness_difference", np.float64),
    ("avg_link_compactness", np.float64), ("avg_link_utilization", np.float64),
    ("fairness", np.float64), ("current_time", np.float64), ("graph_throughput", np.float64),
    ("graph_compactness", np.float64), ("free_total", np.int64), ("obs", np.float64),
]


class Trace(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in TRACE_FIELDS]


def build(asan=False):
    target = "liborlg_oracle_asan.so" if asan else "liborlg_oracle.so"
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)
    return os.path.join(HERE, target)


_LIB = {}


def lib(asan=False):
    if asan not in _LIB:
        path = os.path.join(HERE, "liborlg_oracle_asan.so" if asan else "liborlg_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "orlg_oracle.c")):
            build(asan)
        L = C.CDLL(path)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Topology), C.POINTER(Config), C.c_uint64]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_reseed.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_reset.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_request.argtypes = [C.c_void_p, C.POINTER(Request)]
        L.orc_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(StepResult)]
        L.orc_step_deeprmsa.argtypes = [C.c_void_p, C.c_int, C.POINTER(StepResult)]
        L.orc_policy.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.orc_current_time.restype = C.c_double
        L.orc_current_time.argtypes = [C.c_void_p]
        L.orc_get_available_slots.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_link_stats.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_get_graph_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 3
        L.orc_get_bit_rate_hist.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.orc_num_running.argtypes = [C.c_void_p]
        L.orc_deeprmsa_observation.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_number_slots.argtypes = [C.c_void_p, C.c_int]
        L.orc_is_path_free.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_get_available_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_run.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.POINTER(Trace)]
        L.orc_py_random_stream.argtypes = [C.c_uint64, C.c_int, C.c_void_p]
        L.orc_set_log_fn.argtypes = [C.c_void_p]
        _LIB[asan] = L
    return _LIB[asan]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleEnv:
    """One reference-semantics environment on the CPU.

    ``tables`` is a dict of numpy arrays: pair_path_base, pair_path_count, path_hops, path_se,
    path_length, path_link_off, path_links (int32 / float64) and num_nodes, num_links, k_paths.
    """

    def __init__(self, tables, *, num_slots, episode_length, bit_rates, bit_rate_cum, src_cum, dst_cum,
                 arrival_lambda, holding_lambda, channel_width=12.5, j=1, reward_mode=0, seed=41, asan=False):
        self.L = lib(asan)
        self._keep = []

        def keep(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            self._keep.append(a)
            return a

        t = Topology()
        t.num_nodes, t.num_links, t.k_paths = int(tables["num_nodes"]), int(tables["num_links"]), int(tables["k_paths"])
        t.num_paths = len(tables["path_hops"])
        for name, dt in (("pair_path_base", np.int32), ("pair_path_count", np.int32), ("path_hops", np.int32),
                         ("path_se", np.int32), ("path_length", np.float64), ("path_link_off", np.int32),
                         ("path_links", np.int32)):
            setattr(t, name, _ptr(keep(tables[name], dt)))
        c = Config()
        c.num_slots, c.episode_length, c.num_bit_rates = int(num_slots), int(episode_length), len(bit_rates)
        c.j, c.reward_mode = int(j), int(reward_mode)
        c.arrival_lambda, c.holding_lambda, c.channel_width = float(arrival_lambda), float(holding_lambda), float(channel_width)
        c.bit_rates = _ptr(keep(bit_rates, np.int32))
        c.bit_rate_cum = _ptr(keep(bit_rate_cum, np.float64)) if bit_rate_cum is not None else None   # None: continuous
        c.src_cum = _ptr(keep(src_cum, np.float64))
        c.dst_cum = _ptr(keep(dst_cum, np.float64))
        self.N, self.E, self.K, self.S, self.j = t.num_nodes, t.num_links, t.k_paths, int(num_slots), int(j)
        self.num_bit_rates = len(bit_rates)
        self.obs_dim = 1 + 2 * self.N + (2 * self.j + 3) * self.K
        self._t, self._c = t, c
        self.h = self.L.orc_create(C.byref(t), C.byref(c), C.c_uint64(int(seed)))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seed(self, seed):
        """OpticalNetworkEnv.seed: a fresh random.Random(seed) (optical_network_env.py:266-271) -- the bit-rate draw stays bound to
        the old generator, as in the reference (orc_seed)."""
        self.L.orc_seed(self.h, C.c_uint64(41 if seed is None else seed))

    def reseed(self, seed):
        """Not the reference: a fresh generator for all five draws (the device's orlg_reseed)."""
        self.L.orc_reseed(self.h, C.c_uint64(seed))

    def reset(self, only_episode_counters=True):
        self.L.orc_reset(self.h, 1 if only_episode_counters else 0)

    def request(self):
        r = Request()
        self.L.orc_get_request(self.h, C.byref(r))
        return r

    def step(self, path, slot):
        r = StepResult()
        self.L.orc_step(self.h, int(path), int(slot), C.byref(r))
        return r

    def step_deeprmsa(self, action):
        r = StepResult()
        self.L.orc_step_deeprmsa(self.h, int(action), C.byref(r))
        return r

    def policy(self, name):
        p, s = C.c_int(), C.c_int()
        self.L.orc_policy(self.h, POLICY[name], C.byref(p), C.byref(s))
        return p.value, s.value

    def counters(self):
        c = Counters()
        self.L.orc_get_counters(self.h, C.byref(c))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def current_time(self):
        return self.L.orc_current_time(self.h)

    def available_slots(self):
        a = np.zeros((self.E, self.S), np.uint8)
        self.L.orc_get_available_slots(self.h, _ptr(a))
        return a

    def link_stats(self):
        out = [np.zeros(self.E) for _ in range(4)]
        self.L.orc_get_link_stats(self.h, *[_ptr(a) for a in out])
        return dict(zip(("utilization", "external_fragmentation", "compactness", "last_update"), out))

    def graph_stats(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.L.orc_get_graph_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return {"throughput": a.value, "compactness": b.value, "last_update": c.value}

    def bit_rate_hist(self):
        out = [np.zeros(self.num_bit_rates, np.int64) for _ in range(4)]
        self.L.orc_get_bit_rate_hist(self.h, *[_ptr(a) for a in out])
        return dict(zip(("requested", "provisioned", "episode_requested", "episode_provisioned"), out))

    def num_running(self):
        return self.L.orc_num_running(self.h)

    def observation(self):
        o = np.zeros(self.obs_dim)
        self.L.orc_deeprmsa_observation(self.h, _ptr(o))
        return o

    def simple_matrix_observation(self):
        self.L.orc_simple_matrix_observation.argtypes = [C.c_void_p, C.c_void_p]
        o = np.zeros(2 * self.N + self.E * self.S)
        self.L.orc_simple_matrix_observation(self.h, _ptr(o))
        return o

    def number_slots(self, idp):
        return self.L.orc_get_number_slots(self.h, idp)

    def is_path_free(self, idp, s, n):
        return bool(self.L.orc_is_path_free(self.h, idp, s, n))

    def available_blocks(self, idp):
        st, ln = np.zeros(64, np.int32), np.zeros(64, np.int32)
        n = self.L.orc_get_available_blocks(self.h, idp, _ptr(st), _ptr(ln))
        return st[:n].copy(), ln[:n].copy()

    def run(self, policy, n_steps, reset_on_done=False, actions=None, fields=None, with_obs=False):
        """Returns a dict of per-step arrays (see TRACE_FIELDS). ``fields=[]`` records nothing (timing)."""
        tr = Trace()
        out = {}
        for name, dt in TRACE_FIELDS:
            if name == "obs":
                if with_obs:
                    out[name] = np.zeros((n_steps, self.obs_dim), dt)
                    setattr(tr, name, _ptr(out[name]))
                continue
            if fields is None or name in fields:
                out[name] = np.zeros(n_steps, dt)
                setattr(tr, name, _ptr(out[name]))
        ap = None
        if actions is not None:
            actions = np.ascontiguousarray(actions, np.int32)
            ap = _ptr(actions)
        self.L.orc_run(self.h, POLICY[policy], int(n_steps), 1 if reset_on_done else 0, ap, C.byref(tr))
        return out


def set_log_fn(fn_ptr, asan=False):
    """Use another natural log in expovariate (a C function pointer as int / c_void_p); None = libm."""
    lib(asan).orc_set_log_fn(C.c_void_p(fn_ptr) if fn_ptr else None)


def py_random_stream(seed, n):
    out = np.zeros(n)
    lib().orc_py_random_stream(C.c_uint64(seed), n, _ptr(out))
    return out


# ----------------------------------------------------------------------------------------- PhyRMSA oracle
PHY_MAX_CH = 12
PHY_POLICY = {"bmfa": 0, "bmfa_rss": 1, "sapff": 2, "bmff": 3, "sapbm": 4, "faff": 5, "faff_rss": 6}


class PhyConfig(C.Structure):
    _fields_ = [("num_channels", C.c_int32), ("episode_length", C.c_int32), ("num_bit_rates", C.c_int32),
                ("k_table", C.c_int32), ("grooming", C.c_int32), ("defrag_period", C.c_int32),
                ("number_moves", C.c_int32), ("defrag_metric", C.c_int32),
                ("arrival_lambda", C.c_double), ("holding_lambda", C.c_double)] + \
               [(n, C.c_void_p) for n in ("bit_rates", "bit_rate_cum", "src_cum", "dst_cum", "pair_table_row",
                                          "modulation_level", "gsnr", "link_ends", "path_node_off", "path_nodes")] + \
               [("gn_on", C.c_int32), ("gn_num_thresholds", C.c_int32), ("gn_launch_power_w", C.c_double),
                ("gn_channel_bandwidth_hz", C.c_double), ("gn_attenuation", C.c_double), ("gn_noise_figure", C.c_double),
                ("gn_center_frequency_hz", C.c_void_p), ("gn_link_num_spans", C.c_void_p),
                ("gn_link_span_length_km", C.c_void_p), ("gn_thresholds_db", C.c_void_p)]


class PhyAction(C.Structure):
    _fields_ = [("path", C.c_int32), ("n", C.c_int32), ("ch", C.c_int32 * PHY_MAX_CH), ("cap", C.c_int32 * PHY_MAX_CH),
                ("used", C.c_double * PHY_MAX_CH), ("free_", C.c_double * PHY_MAX_CH)]


class PhyResult(C.Structure):
    _fields_ = [("reward", C.c_double), ("done", C.c_int32), ("accepted", C.c_int32)] + \
               [(n, C.c_double) for n in ("number_cuts_total", "rss_total_metric", "total_path_length", "avrage_gsnr",
                                          "average_path_index", "service_blocking_rate", "episode_service_blocking_rate",
                                          "bit_rate_blocking_rate", "episode_bit_rate_blocking_rate")] + \
               [(n, C.c_int64) for n in ("total_modulation_level", "channels_accepted", "path_index", "physical_paths")] + \
               [("num_moves", C.c_double), ("num_moves_groom", C.c_int64), ("num_defrag_cycle", C.c_int64),
                ("gn_gsnr_db", C.c_double)]


PHY_TRACE_FIELDS = [
    ("service_id", np.int32, 1), ("src", np.int32, 1), ("dst", np.int32, 1), ("bit_rate", np.int32, 1),
    ("act_path", np.int32, 1), ("n_channels", np.int32, 1), ("channels", np.int32, PHY_MAX_CH),
    ("ch_cap", np.int32, PHY_MAX_CH),
    ("arrival", np.float64, 1), ("holding", np.float64, 1), ("ch_used", np.float64, PHY_MAX_CH),
    ("accepted", np.uint8, 1), ("done", np.uint8, 1),
    ("services_accepted", np.int64, 1), ("total_modulation_level", np.int64, 1), ("channels_accepted", np.int64, 1),
    ("path_index", np.int64, 1), ("physical_paths", np.int64, 1), ("n_running", np.int64, 1), ("free_total", np.int64, 1),
    ("number_cuts_total", np.float64, 1), ("rss_total_metric", np.float64, 1), ("total_path_length", np.float64, 1),
    ("avrage_gsnr", np.float64, 1), ("average_path_index", np.float64, 1),
    ("episode_service_blocking_rate", np.float64, 1), ("bit_rate_blocking_rate", np.float64, 1),
    ("current_time", np.float64, 1),
    ("num_moves", np.float64, 1), ("num_moves_groom", np.int64, 1), ("num_defrag_cycle", np.int64, 1),
    ("gn_gsnr_db", np.float64, 1),
]


class PhyTrace(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _, _ in PHY_TRACE_FIELDS]


def _phy_lib(asan=False):
    L = lib(asan)
    if not getattr(L, "_phy_ready", False):
        L.orc_phy_create.restype = C.c_void_p
        L.orc_phy_create.argtypes = [C.POINTER(Topology), C.POINTER(PhyConfig), C.c_uint64]
        L.orc_phy_destroy.argtypes = [C.c_void_p]
        L.orc_phy_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_phy_reseed.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_phy_reset.argtypes = [C.c_void_p, C.c_int]
        L.orc_phy_get_request.argtypes = [C.c_void_p, C.POINTER(Request)]
        L.orc_phy_policy.argtypes = [C.c_void_p, C.c_int, C.POINTER(PhyAction)]
        L.orc_phy_step.argtypes = [C.c_void_p, C.POINTER(PhyAction), C.POINTER(PhyResult)]
        L.orc_phy_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
        L.orc_phy_current_time.restype = C.c_double
        L.orc_phy_current_time.argtypes = [C.c_void_p]
        L.orc_phy_get_available_channels.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_phy_num_running.argtypes = [C.c_void_p]
        L.orc_phy_channel_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_phy_run.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(PhyTrace)]
        L._phy_ready = True
    return L


class PhyOracleEnv:
    """One reference-semantics PhyRMSAEnv (physical + virtual layer, no periodic defragmentation) on the CPU."""

    def __init__(self, tables, *, num_channels, episode_length, bit_rates, bit_rate_cum, src_cum, dst_cum,
                 arrival_lambda, holding_lambda, pair_table_row, modulation_level, gsnr, link_ends, path_node_off,
                 path_nodes, grooming=False, defrag_period=None, number_moves=None, metric="cut", seed=41, asan=False,
                 gn_gate=None):
        self.L = _phy_lib(asan)
        self._keep = []

        def keep(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            self._keep.append(a)
            return _ptr(a)

        t = Topology()
        t.num_nodes, t.num_links, t.k_paths = int(tables["num_nodes"]), int(tables["num_links"]), int(tables["k_paths"])
        t.num_paths = len(tables["path_hops"])
        for name, dt in (("pair_path_base", np.int32), ("pair_path_count", np.int32), ("path_hops", np.int32),
                         ("path_se", np.int32), ("path_length", np.float64), ("path_link_off", np.int32),
                         ("path_links", np.int32)):
            setattr(t, name, keep(tables[name], dt))
        c = PhyConfig()
        c.num_channels, c.episode_length, c.num_bit_rates = int(num_channels), int(episode_length), len(bit_rates)
        c.k_table = int(np.asarray(modulation_level).shape[2])
        c.grooming = 1 if grooming else 0
        c.defrag_period = int(defrag_period or 0)
        c.number_moves = int(number_moves or 0)
        c.defrag_metric = 0 if metric == "cut" else 1
        c.arrival_lambda, c.holding_lambda = float(arrival_lambda), float(holding_lambda)
        c.bit_rates = keep(bit_rates, np.int32)
        c.bit_rate_cum = keep(bit_rate_cum, np.float64)
        c.src_cum = keep(src_cum, np.float64)
        c.dst_cum = keep(dst_cum, np.float64)
        c.pair_table_row = keep(pair_table_row, np.int32)
        c.modulation_level = keep(modulation_level, np.uint8)
        c.gsnr = keep(gsnr, np.float64)
        c.link_ends = keep(link_ends, np.int32)
        c.path_node_off = keep(path_node_off, np.int32)
        c.path_nodes = keep(path_nodes, np.int32)
        if gn_gate is not None:   # GN-model admission check of the chosen channels (not in the reference: orlg_oracle_phy.h)
            c.gn_on = 1
            c.gn_launch_power_w = float(gn_gate["launch_power_w"])
            c.gn_channel_bandwidth_hz = float(gn_gate["channel_bandwidth_hz"])
            c.gn_attenuation = float(gn_gate["attenuation_normalized"])
            c.gn_noise_figure = float(gn_gate["noise_figure"])
            c.gn_center_frequency_hz = keep(gn_gate["channel_center_frequency_hz"], np.float64)
            c.gn_link_num_spans = keep(gn_gate["link_num_spans"], np.int32)
            c.gn_link_span_length_km = keep(gn_gate["link_span_length_km"], np.float64)
            c.gn_thresholds_db = keep(gn_gate["thresholds_db"], np.float64)
            c.gn_num_thresholds = len(gn_gate["thresholds_db"])
        self.E, self.Cn = t.num_links, int(num_channels)
        self._t, self._c = t, c
        self.h = self.L.orc_phy_create(C.byref(t), C.byref(c), C.c_uint64(int(seed)))

    def close(self):
        if self.h:
            self.L.orc_phy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seed(self, seed):
        self.L.orc_phy_seed(self.h, C.c_uint64(41 if seed is None else seed))

    def reseed(self, seed):
        self.L.orc_phy_reseed(self.h, C.c_uint64(seed))

    def reset(self, only_episode_counters=True):
        self.L.orc_phy_reset(self.h, 1 if only_episode_counters else 0)

    def request(self):
        r = Request()
        self.L.orc_phy_get_request(self.h, C.byref(r))
        return r

    def policy(self, name):
        a = PhyAction()
        self.L.orc_phy_policy(self.h, PHY_POLICY[name], C.byref(a))
        return a

    def step(self, action):
        r = PhyResult()
        self.L.orc_phy_step(self.h, C.byref(action), C.byref(r))
        return r

    def counters(self):
        c = Counters()
        self.L.orc_phy_get_counters(self.h, C.byref(c))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def current_time(self):
        return self.L.orc_phy_current_time(self.h)

    def available_channels(self):
        a = np.zeros((self.E, self.Cn), np.uint8)
        self.L.orc_phy_get_available_channels(self.h, _ptr(a))
        return a

    def num_running(self):
        return self.L.orc_phy_num_running(self.h)

    def channel_state(self):
        """env.channel_state as {(src_id, dst_id, k-path): [(channel, used, free, capacity), ...]}, empty lists omitted."""
        out = {}
        buf = np.zeros((256, 4), np.int32)
        N, K = self._t.num_nodes, self._t.k_paths
        for s in range(N):
            for d in range(N):
                for k in range(K):
                    n = self.L.orc_phy_channel_state(self.h, s, d, k, _ptr(buf), 256)
                    if n:
                        out[(s, d, k)] = [tuple(int(x) for x in row) for row in buf[:n]]
        return out

    def run(self, policy, n_steps, reset_on_done=False, fields=None):
        tr = PhyTrace()
        out = {}
        for name, dt, width in PHY_TRACE_FIELDS:
            if fields is None or name in fields:
                out[name] = np.zeros((n_steps, width) if width > 1 else n_steps, dt)
                setattr(tr, name, _ptr(out[name]))
        self.L.orc_phy_run(self.h, PHY_POLICY[policy], int(n_steps), 1 if reset_on_done else 0, C.byref(tr))
        return out


# ----------------------------------------------------------------------------------------- GN-model OSNR oracle
class OsnrBatch(C.Structure):
    _fields_ = [("num_checks", C.c_int32), ("num_links", C.c_int32), ("num_spans", C.c_int32), ("num_services", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("check_link_off", "link_span_off", "link_svc_off", "bandwidth", "center_frequency",
                                          "launch_power", "span_length_km", "span_attenuation", "span_noise_figure",
                                          "svc_bandwidth", "svc_center_frequency", "svc_se", "svc_is_self")]


OSNR_FIELDS = (("check_link_off", np.int32), ("link_span_off", np.int32), ("link_svc_off", np.int32),
               ("bandwidth", np.float64), ("center_frequency", np.float64), ("launch_power", np.float64),
               ("span_length_km", np.float64), ("span_attenuation", np.float64), ("span_noise_figure", np.float64),
               ("svc_bandwidth", np.float64), ("svc_center_frequency", np.float64), ("svc_se", np.int32),
               ("svc_is_self", np.uint8))


def gn_osnr(arrays):
    """GSNR [dB] for a flattened batch of admission checks (dict of arrays, see orlg_oracle_osnr.h)."""
    L = lib()
    L.orc_gn_osnr.argtypes = [C.POINTER(OsnrBatch), C.c_void_p]
    b = OsnrBatch()
    keep = []
    for name, dt in OSNR_FIELDS:
        a = np.ascontiguousarray(arrays[name], dtype=dt)
        keep.append(a)
        setattr(b, name, _ptr(a))
    b.num_checks = len(arrays["bandwidth"])
    b.num_links = len(arrays["link_span_off"]) - 1
    b.num_spans = len(arrays["span_length_km"])
    b.num_services = len(arrays["svc_bandwidth"])
    out = np.zeros(b.num_checks)
    L.orc_gn_osnr(C.byref(b), _ptr(out))
    return out
