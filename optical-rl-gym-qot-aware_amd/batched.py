"""BatchedRMSAEnv: B independent RMSA / DeepRMSA environments stepped on one MI355X.

Constructor kwargs are the reference's (``rmsa_env.py:29-53``, ``deeprmsa_env.py:10-32``) plus
``batch_size`` / ``device`` / ``stats_level`` / ``queue_capacity``.  Environment ``i`` is seeded with
``seed + i`` (or ``seeds[i]``), i.e. it replays ``RMSAEnv(..., seed=seed + i)`` of the reference.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib
from .topology import FrozenTopology, selection_tables

DEFAULT_BIT_RATES = (200, 250, 300, 350, 400, 450, 500, 550, 600, 650, 700, 750, 800, 850, 900, 950, 1000, 1050,
                     1100, 1150, 1200)  # rmsa_env.py:37-38

COUNTER_NAMES = ("services_processed", "services_accepted", "episode_services_processed",
                 "episode_services_accepted", "bit_rate_requested", "bit_rate_provisioned",
                 "episode_bit_rate_requested", "episode_bit_rate_provisioned")

REQUEST_DTYPE = np.dtype([("service_id", np.int32), ("src", np.int32), ("dst", np.int32), ("bit_rate", np.int32),
                          ("arrival_time", np.float64), ("holding_time", np.float64)])


def _ptr(a):
    """Pointer of a numpy array, a torch tensor (host or device) or None."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def _dtype_name(a):
    """'int32', 'float64', ... of a numpy array or a torch tensor."""
    return str(a.dtype).replace("torch.", "")


def _check_buffer(name, a, shape, dtype):
    """A caller-supplied array the library reads or writes through a raw pointer: shape, dtype and layout must be exactly
    what the C ABI expects (a wrong dtype would be reinterpreted, a short array overrun)."""
    if tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(a.shape)}, expected {tuple(shape)}")
    if _dtype_name(a) != str(np.dtype(dtype)):
        raise TypeError(f"{name}: dtype {_dtype_name(a)}, expected {np.dtype(dtype)}")
    contiguous = a.is_contiguous() if hasattr(a, "is_contiguous") else a.flags["C_CONTIGUOUS"]
    if not contiguous:
        raise ValueError(f"{name}: must be C-contiguous")
    if not hasattr(a, "data_ptr") and not a.flags["WRITEABLE"] and name.startswith("out"):
        raise ValueError(f"{name}: read-only array")
    return a


class BatchedRMSAEnv:
    def __init__(self, topology, batch_size: int, *, episode_length: int = 1000, load: float = 10,
                 mean_service_holding_time: float = 10800.0, num_spectrum_resources: int = 100,
                 bit_rate_selection: str = "discrete", bit_rates: Sequence[int] = DEFAULT_BIT_RATES,
                 bit_rate_probabilities=None, node_request_probabilities=None, seed: Optional[int] = None,
                 seeds=None, allow_rejection: bool = False, channel_width: float = 12.5, j: int = 1,
                 reward_mode: int = 0, stats_level: str = "full", queue_capacity: int = 0, device: int = 0,
                 step_kernel: str = "auto", bit_rate_lower_bound=25, bit_rate_higher_bound=100):
        if bit_rate_selection not in ("continuous", "discrete"):   # rmsa_env.py:74
            raise ValueError("bit_rate_selection must be 'continuous' or 'discrete'")
        self.bit_rate_selection = bit_rate_selection
        if bit_rate_selection == "continuous":
            # rmsa_env.py:95-101: rng.randint(lower, higher) -- every integer rate of the range, drawn by rejection on the device
            lo, hi = int(bit_rate_lower_bound), int(bit_rate_higher_bound)
            if lo != bit_rate_lower_bound or hi != bit_rate_higher_bound or hi < lo:
                raise ValueError("bit_rate_lower_bound / bit_rate_higher_bound must be integers, lower <= higher (random.randint)")
            if hi - lo + 1 > 256:
                raise ValueError("continuous bit rates: at most 256 integer rates (lower .. higher)")
            self.bit_rate_lower_bound, self.bit_rate_higher_bound = lo, hi
            bit_rates, bit_rate_probabilities = list(range(lo, hi + 1)), None
        self.L = _lib.load()
        self.topology = FrozenTopology.from_graph(topology)
        t = self.topology
        self.batch_size = int(batch_size)
        self.episode_length = int(episode_length)
        self.num_spectrum_resources = int(num_spectrum_resources)
        self.k_paths = t.k_paths
        self.j = int(j)
        self.allow_rejection = bool(allow_rejection)
        self.reject_action = 1 if allow_rejection else 0
        self.channel_width = float(channel_width)
        self.bit_rates = [int(b) for b in bit_rates]
        # optical_network_env.py:111-129
        self.load = load
        self.mean_service_holding_time = mean_service_holding_time
        self.mean_service_inter_arrival_time = 1 / float(load / float(mean_service_holding_time))
        self.node_request_probabilities, src_cum, dst_cum, br_cum = selection_tables(
            node_request_probabilities, bit_rate_probabilities, t.num_nodes, self.bit_rates)
        self.rand_seed = 41 if seed is None else int(seed)  # optical_network_env.py:266-271
        self.stats_level = stats_level

        self._keep = []

        def keep(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            self._keep.append(a)
            return a.ctypes.data_as(C.c_void_p)

        ct = _lib.Topology()
        ct.num_nodes, ct.num_links, ct.k_paths, ct.num_paths = t.num_nodes, t.num_links, t.k_paths, t.num_paths
        ct.pair_path_base = keep(t.pair_path_base, np.int32)
        ct.pair_path_count = keep(t.pair_path_count, np.int32)
        ct.path_hops = keep(t.path_hops, np.int32)
        ct.path_se = keep(t.path_se, np.int32)
        ct.path_length = keep(t.path_length, np.float64)
        ct.path_link_off = keep(t.path_link_off, np.int32)
        ct.path_links = keep(t.path_links, np.int32)
        cc = _lib.RmsaConfig()
        cc.num_slots, cc.episode_length, cc.num_bit_rates = self.num_spectrum_resources, self.episode_length, len(self.bit_rates)
        cc.j, cc.reward_mode, cc.queue_capacity = self.j, int(reward_mode), int(queue_capacity)
        cc.stats_level = _lib.STATS_LEVELS[stats_level]
        cc.step_kernel = _lib.STEP_KERNELS[step_kernel]
        # rmsa_env.py:646-651: expovariate(1 / mean)
        cc.arrival_lambda = 1 / self.mean_service_inter_arrival_time
        cc.holding_lambda = 1 / self.mean_service_holding_time
        cc.channel_width = self.channel_width
        cc.bit_rates = keep(self.bit_rates, np.int32)
        cc.bit_rate_cum = keep(br_cum, np.float64) if bit_rate_selection == "discrete" else None   # (NULL: rng.randint)
        cc.src_cum = keep(src_cum, np.float64)
        cc.dst_cum = keep(dst_cum, np.float64)
        seeds_ptr = None
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
            assert seeds.shape == (self.batch_size,)
            seeds_ptr = seeds.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        _lib.check(self.L.orlg_create(C.byref(ct), C.byref(cc), self.batch_size, seeds_ptr,
                                      C.c_uint64(self.rand_seed), int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self.words_per_link = self.L.orlg_words_per_link(self.h)
        self.obs_dim = self.L.orlg_deeprmsa_obs_dim(self.h)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "h", None):
            self.L.orlg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        """Run on an existing HIP stream, e.g. ``torch.cuda.current_stream().cuda_stream``."""
        _lib.check(self.L.orlg_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def synchronize(self):
        _lib.check(self.L.orlg_synchronize(self.h))

    def last_kernel(self) -> str:
        """Name, template arguments and launch shape of the kernel behind the last ``run`` / ``reset``."""
        buf = C.create_string_buffer(128)
        _lib.check(self.L.orlg_last_kernel(self.h, buf, 128))
        return buf.value.decode()

    def launch_info(self):
        """Launch geometry of the step kernel (envs per workgroup, LDS bytes, resident workgroups per CU)."""
        a = np.zeros(4, np.int32)
        _lib.check(self.L.orlg_launch_info(self.h, _ptr(a)))
        return {"envs_per_workgroup": int(a[0]), "lds_bytes_per_workgroup": int(a[1]),
                "workgroups_per_cu": int(a[2]), "words_per_link": int(a[3])}

    # ------------------------------------------------------------------ stepping
    def reset(self, only_episode_counters: bool = True):
        _lib.check(self.L.orlg_reset(self.h, 1 if only_episode_counters else 0))

    def reseed(self, seed=None, seeds=None):
        """A fresh ``random.Random`` for every environment -- ``seeds[i]`` if given, else ``seed + i`` (the constructor's
        convention) -- and nothing else changes: the pending requests stay, the next arrivals are the new generators' first
        draws.  NOT the reference's ``seed()``: there the bit-rate draw stays bound to the generator object of construction time
        (``functools.partial(self.rng.choices, ...)``, ``rmsa_env.py:109-111``), so after ``env.seed(s)`` the reference takes
        inter-arrival time, holding time, source and destination from ``Random(s)`` and the bit rate from the OLD generator
        (pinned by ``tests/golden/seed_rmsa_nsfnet_s10.npz``); here all five draws come from the new one."""
        if seeds is not None:
            sa = np.ascontiguousarray(seeds, np.uint64)
            if sa.shape != (self.batch_size,):
                raise ValueError(f"seeds: shape {sa.shape}, expected ({self.batch_size},)")
            _lib.check(self.L.orlg_reseed(self.h, _ptr(sa), 0))
        else:
            _lib.check(self.L.orlg_reseed(self.h, None, int(41 if seed is None else seed)))

    def run(self, policy: str, n_steps: int = 1, *, actions=None, auto_reset: bool = False,
            outputs: Sequence[str] = (), out: Optional[Dict[str, object]] = None):
        """``n_steps`` x (policy -> step) on the device.  ``outputs`` names per-step arrays to return
        (see ``_lib.STEP_IO_DTYPES``) as numpy arrays of shape [n_steps, B(, 4)]; ``out`` may supply
        preallocated numpy arrays or torch tensors (device tensors are written without staging)."""
        B = self.batch_size
        io = _lib.StepIO()
        res = {}
        names = list(outputs) + [k for k in (out or {}) if k not in outputs]
        for name in names:
            if name not in _lib.STEP_IO_DTYPES:
                raise KeyError(f"unknown step output {name!r}")
            shape = (n_steps, B, 4) if name == "request" else (n_steps, B)
            if out is not None and name in out:
                arr = _check_buffer(f"out[{name!r}]", out[name], shape, _lib.STEP_IO_DTYPES[name])
            else:
                arr = np.zeros(shape, dtype=_lib.STEP_IO_DTYPES[name])
            res[name] = arr
            setattr(io, name, _ptr(arr))
        ap = None
        if policy in ("external", "deeprmsa_external", "path_ff_external"):
            if actions is None:
                raise ValueError(f"policy {policy!r} needs an actions array")
            ashape = (B, 2) if policy == "external" else (B,)
            if not hasattr(actions, "data_ptr"):
                actions = np.asarray(actions)
                if actions.dtype.kind not in "iu":
                    raise TypeError(f"actions: dtype {actions.dtype}, expected an integer type")
                actions = np.ascontiguousarray(actions, dtype=np.int32)
            _check_buffer("actions", actions, ashape, np.int32)
            ap = _ptr(actions)
        _lib.check(self.L.orlg_step(self.h, _lib.POLICIES[policy], int(n_steps), ap, 1 if auto_reset else 0,
                                    C.byref(io)))
        return res

    def step(self, actions, outputs=("reward", "done", "accepted")):
        """RMSAEnv.step for every env: ``actions`` is [B, 2] int32 (path, initial_slot)."""
        r = self.run("external", 1, actions=actions, outputs=outputs)
        return {k: v[0] for k, v in r.items()}

    def step_deeprmsa(self, actions, outputs=("reward", "done", "accepted")):
        """DeepRMSAEnv.step for every env: ``actions`` is [B] int32 in Discrete(k*j + reject)."""
        r = self.run("deeprmsa_external", 1, actions=actions, outputs=outputs)
        return {k: v[0] for k, v in r.items()}

    # ------------------------------------------------------------------ state read-back
    def requests(self):
        a = np.zeros(self.batch_size, REQUEST_DTYPE)
        _lib.check(self.L.orlg_get_requests(self.h, _ptr(a)))
        return a

    def counters(self):
        a = np.zeros((self.batch_size, 8), np.int64)
        _lib.check(self.L.orlg_get_counters(self.h, _ptr(a)))
        return {n: a[:, i].copy() for i, n in enumerate(COUNTER_NAMES)}

    def current_time(self):
        a = np.zeros(self.batch_size, np.float64)
        _lib.check(self.L.orlg_get_current_time(self.h, _ptr(a)))
        return a

    def occupancy_words(self):
        E, W = self.topology.num_links, self.words_per_link
        a = np.zeros((self.batch_size, E, W), np.uint64)
        _lib.check(self.L.orlg_get_occupancy(self.h, _ptr(a)))
        return a

    def available_slots(self):
        """topology.graph["available_slots"] for every env: [B, E, S] uint8 (1 = free)."""
        w = self.occupancy_words()
        bits = np.unpackbits(w.view(np.uint8), axis=-1, bitorder="little")
        return bits.reshape(self.batch_size, self.topology.num_links, -1)[:, :, :self.num_spectrum_resources]

    def link_stats(self):
        B, E = self.batch_size, self.topology.num_links
        out = [np.zeros((B, E)) for _ in range(4)]
        _lib.check(self.L.orlg_get_link_stats(self.h, *[_ptr(a) for a in out]))
        return dict(zip(("utilization", "external_fragmentation", "compactness", "last_update"), out))

    def graph_stats(self):
        out = [np.zeros(self.batch_size) for _ in range(3)]
        _lib.check(self.L.orlg_get_graph_stats(self.h, *[_ptr(a) for a in out]))
        return dict(zip(("throughput", "compactness", "last_update"), out))

    def bit_rate_hist(self):
        B, n = self.batch_size, len(self.bit_rates)
        out = [np.zeros((B, n), np.int64) for _ in range(4)]
        _lib.check(self.L.orlg_get_bit_rate_hist(self.h, *[_ptr(a) for a in out]))
        return dict(zip(("requested", "provisioned", "episode_requested", "episode_provisioned"), out))

    def num_running(self):
        a = np.zeros(self.batch_size, np.int32)
        _lib.check(self.L.orlg_get_num_running(self.h, _ptr(a)))
        return a

    def episodes_done(self):
        a = np.zeros(self.batch_size, np.int64)
        _lib.check(self.L.orlg_get_episodes_done(self.h, _ptr(a)))
        return a

    def path_masks(self, env_index: int = 0):
        """(masks [k, W] uint64, nslots [k]) for the pending request of one env."""
        m = np.zeros((self.k_paths, self.words_per_link), np.uint64)
        n = np.zeros(self.k_paths, np.int32)
        _lib.check(self.L.orlg_query_path_masks(self.h, int(env_index), _ptr(m), _ptr(n)))
        return m, n

    def path_mask(self, path_gid: int, env_index: int = 0):
        """Free bitmap [W] of one arbitrary path record and its slot demand for the pending bit rate."""
        m = np.zeros(self.words_per_link, np.uint64)
        n = np.zeros(1, np.int32)
        _lib.check(self.L.orlg_query_path_mask(self.h, int(env_index), int(path_gid), _ptr(m), _ptr(n)))
        return m, int(n[0])

    def observation(self, out=None, dtype=np.float64):
        """DeepRMSAEnv.observation() for every env: [B, obs_dim] float64 (the reference's Box dtype), or float32 -- the
        float64 vector rounded once, ``obs.astype(np.float32)`` -- with ``dtype=np.float32`` / a float32 ``out`` buffer."""
        if out is None:
            out = np.zeros((self.batch_size, self.obs_dim), dtype)
        elif _dtype_name(out) == "float32":
            dtype = np.float32
        if np.dtype(dtype) == np.float32:
            _check_buffer("out", out, (self.batch_size, self.obs_dim), np.float32)
            _lib.check(self.L.orlg_deeprmsa_observation_f32(self.h, _ptr(out)))
            return out
        _check_buffer("out", out, (self.batch_size, self.obs_dim), np.float64)
        _lib.check(self.L.orlg_deeprmsa_observation(self.h, _ptr(out)))
        return out

    def simple_matrix_observation(self, out=None):
        """SimpleMatrixObservation.observation() for every env: [B, 2N + E*S] uint8 (``rmsa_env.py:940-971``)."""
        dim = self.L.orlg_simple_matrix_obs_dim(self.h)
        if out is None:
            out = np.zeros((self.batch_size, dim), np.uint8)
        else:
            _check_buffer("out", out, (self.batch_size, dim), np.uint8)
        _lib.check(self.L.orlg_simple_matrix_observation(self.h, _ptr(out)))
        return out

    def step_path_first_fit(self, paths, outputs=("reward", "done", "accepted")):
        """PathOnlyFirstFitAction.step for every env: ``paths`` is [B] int32 (k = reject)."""
        r = self.run("path_ff_external", 1, actions=paths, outputs=outputs)
        return {k: v[0] for k, v in r.items()}

    def save_state(self):
        """Snapshot of the complete simulation state of the batch (a uint8 array): checkpoint / resume, env cloning."""
        n = self.L.orlg_state_size(self.h)
        if n < 0:
            _lib.check(int(n))
        buf = np.empty(int(n), np.uint8)
        _lib.check(self.L.orlg_save_state(self.h, _ptr(buf)))
        return buf

    def load_state(self, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        assert buf.size == self.L.orlg_state_size(self.h), "snapshot of a differently configured batch"
        _lib.check(self.L.orlg_load_state(self.h, _ptr(buf)))

    def reduce_counters(self):
        """Summed counters of this shard (raises if a release queue overflowed): the vector a
        multi-GPU job all-reduces."""
        a = np.zeros(16, np.int64)
        _lib.check(self.L.orlg_reduce_counters(self.h, _ptr(a)))
        d = {n: int(a[i]) for i, n in enumerate(COUNTER_NAMES)}
        d["episodes_done"], d["num_envs"] = int(a[8]), int(a[9])
        return d, a


class BatchedDeepRMSAEnv(BatchedRMSAEnv):
    """B x DeepRMSAEnv (``deeprmsa_env.py:9-46``): load = holding / inter-arrival, reward +1/-1, actions in
    Discrete(k*j + reject), observation = per-path free-block features."""

    def __init__(self, topology, batch_size: int, *, j: int = 1, episode_length: int = 1000,
                 mean_service_holding_time: float = 25.0, mean_service_inter_arrival_time: float = 0.1,
                 num_spectrum_resources: int = 100, node_request_probabilities=None, seed=None, seeds=None,
                 allow_rejection: bool = False, **extra):
        super().__init__(topology, batch_size, episode_length=episode_length,
                         load=mean_service_holding_time / mean_service_inter_arrival_time,
                         mean_service_holding_time=mean_service_holding_time,
                         num_spectrum_resources=num_spectrum_resources,
                         node_request_probabilities=node_request_probabilities, seed=seed, seeds=seeds,
                         allow_rejection=allow_rejection, j=j, reward_mode=1, **extra)

    def step(self, actions, outputs=("reward", "done", "accepted")):
        return self.step_deeprmsa(actions, outputs=outputs)
