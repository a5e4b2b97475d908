"""ctypes binding of liborlg.so (include/orlg.h).  The library is built in-tree by build.py; there is
no Python or CPU fallback -- if the shared object is missing it is built, and if no HIP device is
visible ``orlg_create`` fails and :class:`OrlgError` is raised."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORLG_LIB_PATH") or os.path.join(HERE, "liborlg.so")  # ORLG_LIB_PATH: debug builds (tools/)

ORLG_OK = 0
ERR_NAMES = {-1: "ORLG_ERR_INVALID", -2: "ORLG_ERR_NO_DEVICE", -3: "ORLG_ERR_HIP", -4: "ORLG_ERR_QUEUE_FULL"}

STATS_LEVELS = {"counters": 0, "network": 1, "full": 2}
POLICIES = {"external": -1, "sp_ff": 0, "sap_ff": 1, "llp_ff": 2, "deeprmsa_sp_ff": 3, "deeprmsa_sap_ff": 4,
            "deeprmsa_external": 5, "path_ff_external": 6}


class OrlgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class Topology(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("num_links", C.c_int32), ("k_paths", C.c_int32), ("num_paths", C.c_int32),
                ("pair_path_base", C.c_void_p), ("pair_path_count", C.c_void_p), ("path_hops", C.c_void_p),
                ("path_se", C.c_void_p), ("path_length", C.c_void_p), ("path_link_off", C.c_void_p),
                ("path_links", C.c_void_p)]


STEP_KERNELS = {"auto": 0, "wave": 1, "group": 2}   # include/orlg.h ORLG_KERNEL_*


class RmsaConfig(C.Structure):
    _fields_ = [("num_slots", C.c_int32), ("episode_length", C.c_int32), ("num_bit_rates", C.c_int32),
                ("j", C.c_int32), ("reward_mode", C.c_int32), ("queue_capacity", C.c_int32),
                ("stats_level", C.c_int32), ("step_kernel", C.c_int32),
                ("arrival_lambda", C.c_double), ("holding_lambda", C.c_double), ("channel_width", C.c_double),
                ("bit_rates", C.c_void_p), ("bit_rate_cum", C.c_void_p), ("src_cum", C.c_void_p),
                ("dst_cum", C.c_void_p)]


class StepIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("act_path", "act_slot", "accepted", "done", "reward", "request", "arrival",
                                         "holding", "network_compactness", "network_compactness_difference",
                                         "avg_link_compactness", "avg_link_utilization")]


# numpy dtypes of the step outputs
STEP_IO_DTYPES = {"act_path": "int32", "act_slot": "int32", "accepted": "uint8", "done": "uint8", "reward": "float64",
                  "request": "int32", "arrival": "float64", "holding": "float64", "network_compactness": "float64",
                  "network_compactness_difference": "float64", "avg_link_compactness": "float64",
                  "avg_link_utilization": "float64"}

class PhyConfig(C.Structure):
    _fields_ = [("num_channels", C.c_int32), ("episode_length", C.c_int32), ("num_bit_rates", C.c_int32),
                ("k_table", C.c_int32), ("num_table_rows", C.c_int32), ("queue_capacity", C.c_int32),
                ("grooming", C.c_int32), ("channel_state_capacity", C.c_int32),
                ("defrag_period", C.c_int32), ("number_moves", C.c_int32), ("defrag_metric", C.c_int32),
                ("defrag_capacity", C.c_int32),
                ("arrival_lambda", C.c_double), ("holding_lambda", C.c_double)] + \
               [(n, C.c_void_p) for n in ("bit_rates", "bit_rate_cum", "src_cum", "dst_cum", "pair_table_row",
                                          "modulation_level", "gsnr", "adj_off", "adj_link", "adj_weight",
                                          "path_node_weights", "node_degree", "link_ends", "gn_gate")]


class GnGate(C.Structure):
    _fields_ = [("launch_power_w", C.c_double), ("channel_bandwidth_hz", C.c_double), ("attenuation_normalized", C.c_double),
                ("noise_figure", C.c_double), ("channel_center_frequency_hz", C.c_void_p), ("link_num_spans", C.c_void_p),
                ("link_span_length_km", C.c_void_p), ("thresholds_db", C.c_void_p), ("num_thresholds", C.c_int32),
                ("pad", C.c_int32)]


class PhyStepIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("act_path", "n_channels", "channels", "accepted", "done", "request",
                                         "arrival", "holding", "number_cuts_total", "rss_total_metric",
                                         "channels_used", "defrag_counters", "gn_gsnr_db")]


PHY_MAX_CHANNELS = 14
PHY_STEP_IO_DTYPES = {"act_path": "int32", "n_channels": "int32", "channels": "int16", "accepted": "uint8",
                      "done": "uint8", "request": "int32", "arrival": "float64", "holding": "float64",
                      "number_cuts_total": "float64", "rss_total_metric": "float64", "channels_used": "int16",
                      "defrag_counters": "int32", "gn_gsnr_db": "float64"}
PHY_POLICIES = {"external": -1, "bmfa": 0, "bmfa_rss": 1, "sapff": 2, "bmff": 3, "sapbm": 4, "faff": 5, "faff_rss": 6}

_lib = None


def load(build_if_missing=True):
    """Load liborlg.so, building it with hipcc first when it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and not os.environ.get("ORLG_LIB_PATH"):
        from . import build as _build
        if _build.needs_build():
            _build.build(verbose=False)
    if not os.path.exists(LIB_PATH):
        raise OrlgError(-2, f"{LIB_PATH} is missing: run python optical-rl-gym-qot-aware_amd/build.py (needs hipcc)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    L.orlg_abi_version.restype = C.c_int
    L.orlg_last_error.restype = C.c_char_p
    L.orlg_device_count.restype = C.c_int
    L.orlg_create.argtypes = [C.POINTER(Topology), C.POINTER(RmsaConfig), i32, vp, u64, i32, C.POINTER(vp)]
    L.orlg_destroy.argtypes = [vp]
    L.orlg_set_stream.argtypes = [vp, vp]
    L.orlg_synchronize.argtypes = [vp]
    L.orlg_launch_info.argtypes = [vp, vp]
    L.orlg_last_kernel.argtypes = [vp, C.c_char_p, i32]
    L.orlg_phy_last_kernel.argtypes = [vp, C.c_char_p, i32]
    L.orlg_reset.argtypes = [vp, i32]
    L.orlg_reseed.argtypes = [vp, vp, C.c_uint64]
    L.orlg_step.argtypes = [vp, i32, i32, vp, i32, C.POINTER(StepIO)]
    L.orlg_get_requests.argtypes = [vp, vp]
    L.orlg_get_counters.argtypes = [vp, vp]
    L.orlg_get_current_time.argtypes = [vp, vp]
    L.orlg_get_occupancy.argtypes = [vp, vp]
    L.orlg_words_per_link.argtypes = [vp]
    L.orlg_get_link_stats.argtypes = [vp, vp, vp, vp, vp]
    L.orlg_get_graph_stats.argtypes = [vp, vp, vp, vp]
    L.orlg_get_bit_rate_hist.argtypes = [vp, vp, vp, vp, vp]
    L.orlg_get_num_running.argtypes = [vp, vp]
    L.orlg_get_episodes_done.argtypes = [vp, vp]
    L.orlg_query_path_masks.argtypes = [vp, i32, vp, vp]
    L.orlg_query_path_mask.argtypes = [vp, i32, i32, vp, vp]
    L.orlg_deeprmsa_observation.argtypes = [vp, vp]
    L.orlg_deeprmsa_observation_f32.argtypes = [vp, vp]
    L.orlg_deeprmsa_obs_dim.argtypes = [vp]
    L.orlg_reduce_counters.argtypes = [vp, vp]
    L.orlg_simple_matrix_observation.argtypes = [vp, vp]
    L.orlg_simple_matrix_obs_dim.argtypes = [vp]
    L.orlg_phy_create.argtypes = [C.POINTER(Topology), C.POINTER(PhyConfig), i32, vp, u64, i32, C.POINTER(vp)]
    L.orlg_phy_destroy.argtypes = [vp]
    L.orlg_phy_set_stream.argtypes = [vp, vp]
    L.orlg_phy_synchronize.argtypes = [vp]
    L.orlg_phy_reset.argtypes = [vp, i32]
    L.orlg_phy_reseed.argtypes = [vp, vp, C.c_uint64]
    L.orlg_phy_step.argtypes = [vp, i32, i32, vp, vp, i32, C.POINTER(PhyStepIO)]
    L.orlg_phy_words_per_link.argtypes = [vp]
    L.orlg_phy_node_vectors.argtypes = [vp]
    for name in ("orlg_phy_get_requests", "orlg_phy_get_counters", "orlg_phy_get_current_time",
                 "orlg_phy_get_num_running", "orlg_phy_get_episode_stats", "orlg_phy_get_occupancy",
                 "orlg_phy_reduce_counters"):
        getattr(L, name).argtypes = [vp, vp]
    for name in ("orlg_state_size", "orlg_phy_state_size"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = i64
    for name in ("orlg_save_state", "orlg_load_state", "orlg_phy_save_state", "orlg_phy_load_state"):
        getattr(L, name).argtypes = [vp, vp]
    L.orlg_phy_get_channel_state.argtypes = [vp, i32, vp, vp]
    L.orlg_phy_channel_state_capacity.argtypes = [vp]
    L.orlg_host_log.argtypes = [C.c_double]
    L.orlg_host_log.restype = C.c_double
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "orlg_abi_version", "orlg_last_error", "orlg_device_count", "orlg_create", "orlg_destroy", "orlg_set_stream",
    "orlg_synchronize", "orlg_launch_info", "orlg_last_kernel", "orlg_phy_last_kernel", "orlg_reset", "orlg_reseed", "orlg_step", "orlg_get_requests", "orlg_get_counters",
    "orlg_get_current_time", "orlg_get_occupancy", "orlg_words_per_link", "orlg_get_link_stats",
    "orlg_get_graph_stats", "orlg_get_bit_rate_hist", "orlg_get_num_running", "orlg_get_episodes_done",
    "orlg_query_path_masks", "orlg_query_path_mask", "orlg_deeprmsa_observation", "orlg_deeprmsa_observation_f32", "orlg_deeprmsa_obs_dim", "orlg_reduce_counters",
    "orlg_simple_matrix_observation", "orlg_simple_matrix_obs_dim", "orlg_host_log",
    "orlg_phy_create", "orlg_phy_destroy", "orlg_phy_set_stream", "orlg_phy_synchronize", "orlg_phy_reset", "orlg_phy_reseed",
    "orlg_phy_step", "orlg_phy_words_per_link", "orlg_phy_node_vectors", "orlg_phy_get_requests", "orlg_phy_get_counters",
    "orlg_phy_get_current_time", "orlg_phy_get_num_running", "orlg_phy_get_episode_stats",
    "orlg_phy_get_occupancy", "orlg_phy_reduce_counters", "orlg_phy_get_channel_state",
    "orlg_phy_channel_state_capacity", "orlg_gn_osnr", "orlg_state_size", "orlg_save_state", "orlg_load_state",
    "orlg_phy_state_size", "orlg_phy_save_state", "orlg_phy_load_state",
]


def check(rc):
    if rc != ORLG_OK:
        raise OrlgError(rc, load().orlg_last_error().decode())
