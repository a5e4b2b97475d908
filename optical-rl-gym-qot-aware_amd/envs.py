"""Single-environment gym-style views with the reference's object surface.

``RMSAEnv`` / ``DeepRMSAEnv`` here are drop-ins for ``optical_rl_gym.envs.rmsa_env.RMSAEnv`` and
``optical_rl_gym.envs.deeprmsa_env.DeepRMSAEnv``: same constructor kwargs, ``reset(only_episode_counters=True)``,
``step(action) -> (obs, reward, done, info)`` with the same ``info`` keys, and the attributes / query methods the
reference's heuristic callbacks ``f(env) -> action`` touch (``rmsa_env.py:854-937``, ``deeprmsa_env.py:135-155``).
The environment itself lives on the GPU (a :class:`BatchedRMSAEnv` of batch 1, or one row of a bigger batch);
every query is a read of device state through the C ABI (``orlg_query_path_masks`` computes the path-wide free
bitmaps on the device, this module only slices them).  There is no CPU simulation here.

The heuristics at the bottom are this package's own statements of the reference's policies on that surface;
the reference's functions run unchanged on these classes as well (same names, same attributes).
"""
from __future__ import annotations

import math
from collections import defaultdict
from typing import Optional, Sequence, Tuple

import numpy as np

from .batched import DEFAULT_BIT_RATES, BatchedDeepRMSAEnv, BatchedRMSAEnv
from .topology import FrozenTopology, Path, Service

try:  # gym is optional: the reference needs it, this package only mirrors its spaces when present
    import gym as _gym  # type: ignore
except Exception:  # pragma: no cover - gym is not installed in the build image
    try:
        import gymnasium as _gym  # type: ignore
    except Exception:
        _gym = None


class _Space:
    """Minimal stand-in used when neither gym nor gymnasium is importable."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        if self.kind == "MultiDiscrete":
            return tuple(int(self._rng.integers(0, n)) for n in self.nvec)
        if self.kind == "Discrete":
            return int(self._rng.integers(0, self.n))
        raise NotImplementedError(self.kind)


def _multi_discrete(nvec):
    if _gym is not None:
        return _gym.spaces.MultiDiscrete(nvec)
    return _Space("MultiDiscrete", nvec=tuple(nvec))


def _discrete(n):
    if _gym is not None:
        return _gym.spaces.Discrete(n)
    return _Space("Discrete", n=n)


def _box(low, high, shape, dtype):
    if _gym is not None:
        return _gym.spaces.Box(low=low, high=high, dtype=dtype, shape=shape)
    return _Space("Box", low=low, high=high, shape=shape, dtype=dtype)


class _LazyGraph(dict):
    """``topology.graph``: static keys plus device-backed ones fetched on access."""

    def __init__(self, static, env):
        super().__init__(static)
        self._env = env

    def __getitem__(self, key):
        if key == "available_slots":
            return self._env._available_slots()
        if key in ("throughput", "compactness", "last_update"):
            return float(self._env._batched.graph_stats()[key][self._env._index])
        return super().__getitem__(key)


class RMSAEnv:
    """Drop-in for ``RMSAEnv`` (``rmsa_env.py:18``) backed by the device path."""

    metadata = {"metrics": ["service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate",
                            "episode_bit_rate_blocking_rate"]}
    _batched_cls = BatchedRMSAEnv

    def __init__(self, topology=None, episode_length: int = 1000, load: float = 10,
                 mean_service_holding_time: float = 10800.0, num_spectrum_resources: int = 100,
                 bit_rate_selection: str = "discrete", bit_rates: Sequence = DEFAULT_BIT_RATES,
                 bit_rate_probabilities=None, node_request_probabilities=None, bit_rate_lower_bound: float = 25.0,
                 bit_rate_higher_bound: float = 100.0, seed: Optional[int] = None, allow_rejection: bool = False,
                 reset: bool = True, channel_width: float = 12.5, device: int = 0, _batched=None, _index: int = 0,
                 **_ignored):
        assert bit_rate_selection in ("continuous", "discrete")
        if _batched is None:
            _batched = self._batched_cls(topology, 1, episode_length=episode_length, load=load,
                                         mean_service_holding_time=mean_service_holding_time,
                                         num_spectrum_resources=num_spectrum_resources,
                                         bit_rate_selection=bit_rate_selection, bit_rates=bit_rates,
                                         bit_rate_lower_bound=bit_rate_lower_bound, bit_rate_higher_bound=bit_rate_higher_bound,
                                         bit_rate_probabilities=bit_rate_probabilities,
                                         node_request_probabilities=node_request_probabilities, seed=seed,
                                         allow_rejection=allow_rejection, channel_width=channel_width, device=device)
        self._init_view(_batched, _index)

    # ------------------------------------------------------------------ plumbing
    def _init_view(self, batched, index):
        self._batched = batched
        self._index = int(index)
        ft: FrozenTopology = batched.topology
        self._ft = ft
        view = ft.view()
        view.graph = _LazyGraph(dict(view.graph, num_spectrum_resources=batched.num_spectrum_resources), self)
        self.topology = view
        self.topology_name = ft.name
        self.k_paths = ft.k_paths
        self.k_shortest_paths = ft.ksp
        self.num_spectrum_resources = batched.num_spectrum_resources
        self.episode_length = batched.episode_length
        self.channel_width = batched.channel_width
        self.allow_rejection = batched.allow_rejection
        self.reject_action = batched.reject_action
        self.bit_rate_selection = getattr(batched, "bit_rate_selection", "discrete")
        if self.bit_rate_selection == "discrete":   # (rmsa_env.py:103-111: the attribute exists in this mode only)
            self.bit_rates = list(batched.bit_rates)
        else:
            self.bit_rate_lower_bound, self.bit_rate_higher_bound = batched.bit_rate_lower_bound, batched.bit_rate_higher_bound
        self.load = batched.load
        self.mean_service_holding_time = batched.mean_service_holding_time
        self.mean_service_inter_arrival_time = batched.mean_service_inter_arrival_time
        self.node_request_probabilities = batched.node_request_probabilities
        self.rand_seed = batched.rand_seed + self._index
        self.j = batched.j
        self.action_space = _multi_discrete((self.k_paths + self.reject_action,
                                             self.num_spectrum_resources + self.reject_action))
        self.observation_space = None
        self.action_space.seed(self.rand_seed)
        self.current_service: Optional[Service] = None
        # bookkeeping the reference keeps beside the simulation (no info key reads it): kept here, on the host, from what a step
        # returns -- rmsa_env.py:185-196 (shape (k + 1, S + 1) at construction), :118-128
        shape = (self.k_paths + 1, self.num_spectrum_resources + 1)
        self.actions_output = np.zeros(shape, dtype=int)
        self.episode_actions_output = np.zeros(shape, dtype=int)
        self.actions_taken = np.zeros(shape, dtype=int)
        self.episode_actions_taken = np.zeros(shape, dtype=int)
        if self.bit_rate_selection == "discrete":
            self.slots_requested_histogram = defaultdict(int)
            self.episode_slots_requested_histogram = defaultdict(int)
            self.slots_provisioned_histogram = defaultdict(int)
            self.episode_slots_provisioned_histogram = defaultdict(int)
        self._sync()
        self._count_request()

    def _count_request(self):
        """``_next_service`` (rmsa_env.py:676-686): the histogram of slots requested, assuming the shortest path."""
        if self.bit_rate_selection == "discrete":
            slots = self.get_number_slots(self._candidates[0])
            self.slots_requested_histogram[slots] += 1
            self.episode_slots_requested_histogram[slots] += 1

    def _reset_bookkeeping(self, only_episode_counters):
        """``reset`` (rmsa_env.py:343-389): the episode arrays are re-created (with the reject action's shape), the pending
        request counted again; the arrays of the whole run are never cleared."""
        shape = (self.k_paths + self.reject_action, self.num_spectrum_resources + self.reject_action)
        self.episode_actions_output = np.zeros(shape, dtype=int)
        self.episode_actions_taken = np.zeros(shape, dtype=int)
        if self.bit_rate_selection == "discrete":
            self.episode_slots_requested_histogram = defaultdict(int)
            self.episode_slots_provisioned_histogram = defaultdict(int)
            if only_episode_counters:
                self.episode_slots_requested_histogram[self.get_number_slots(self._candidates[0])] += 1
        if not only_episode_counters:
            self._count_request()   # (the full reset ends with _next_service)

    def _sync(self):
        """Refresh the host mirror of the pending request, the counters and the per-path masks."""
        b, i = self._batched, self._index
        r = b.requests()[i]
        nodes = self._ft.nodes
        self.current_service = Service(int(r["service_id"]), nodes[r["src"]], int(r["src"]),
                                       destination=nodes[r["dst"]], destination_id=int(r["dst"]),
                                       arrival_time=float(r["arrival_time"]), holding_time=float(r["holding_time"]),
                                       bit_rate=int(r["bit_rate"]))
        c = b.counters()
        for name, arr in c.items():
            setattr(self, name, int(arr[i]))
        self.current_time = float(b.current_time()[i])
        masks, nslots = b.path_masks(i)
        bits = np.unpackbits(masks.view(np.uint8), axis=-1, bitorder="little")[:, :self.num_spectrum_resources]
        self._path_bits = bits.astype(np.int64)
        self._path_nslots = nslots
        self._candidates = self.k_shortest_paths[self.current_service.source, self.current_service.destination]
        self._avail_cache = None

    def _available_slots(self):
        if self._avail_cache is None:
            self._avail_cache = self._batched.available_slots()[self._index].astype(np.int64)
        return self._avail_cache

    def _bits_of(self, path: Path):
        for idp, p in enumerate(self._candidates):
            if p is path:
                return self._path_bits[idp]
        m, _ = self._batched.path_mask(path.gid, self._index)
        return np.unpackbits(m.view(np.uint8), bitorder="little")[:self.num_spectrum_resources].astype(np.int64)

    # ------------------------------------------------------------------ reference query surface
    def get_number_slots(self, path: Path) -> int:
        """``rmsa_env.py:708-719`` (plain arithmetic on the pending request)."""
        return math.ceil(self.current_service.bit_rate /
                         (path.best_modulation.spectral_efficiency * self.channel_width)) + 1

    def is_path_free(self, path: Path, initial_slot: int, number_slots: int) -> bool:
        """``rmsa_env.py:721-734``"""
        if initial_slot + number_slots > self.num_spectrum_resources:
            return False
        return bool(self._bits_of(path)[initial_slot:initial_slot + number_slots].all())

    def get_available_slots(self, path: Path):
        """``rmsa_env.py:745-756``"""
        return self._bits_of(path).copy()

    @staticmethod
    def rle(inarray):
        """``rmsa_env.py:758-772``: (start positions, values, run lengths)."""
        ia = np.asarray(inarray)
        n = len(ia)
        if n == 0:
            return None, None, None
        change = np.flatnonzero(ia[1:] != ia[:-1])
        ends = np.append(change, n - 1)
        lengths = np.diff(np.append(-1, ends))
        starts = np.cumsum(np.append(0, lengths))[:-1]
        return starts, ia[ends], lengths

    def get_available_blocks(self, path: int):
        """``rmsa_env.py:774-804``: first j free runs long enough for the pending request on candidate ``path``."""
        cand = self._candidates[path]
        slots = self.get_number_slots(cand)
        starts, values, lengths = self.rle(self._path_bits[path])
        ok = np.flatnonzero((values == 1) & (lengths >= slots))[: self.j]
        return starts[ok], lengths[ok]

    # ------------------------------------------------------------------ gym surface
    def observation(self):
        return {"topology": self.topology, "current_service": self.current_service}

    def reward(self):
        return 1 if self.current_service.accepted else 0

    def seed(self, seed=None):
        """``optical_network_env.py:266-271``; see ``BatchedRMSAEnv.reseed`` for why this is refused."""
        raise NotImplementedError(
            "seed() after construction is not reproduced: the reference keeps drawing the BIT RATE from the generator object of "
            "construction time (functools.partial(self.rng.choices, ...)) while the other four draws of a request come from "
            "Random(seed) -- two generators per environment.  Pass seed= to the constructor, or call reseed() on the batched "
            "environment for a fresh generator for all draws (not the reference's stream).")

    def render(self, mode="human"):
        return

    def reset(self, only_episode_counters: bool = True):
        """``rmsa_env.py:343-457``"""
        if self._batched.batch_size != 1:
            raise RuntimeError("reset() of a view into a larger batch would reset every env: use the batched API")
        self._batched.reset(only_episode_counters)
        self._sync()
        self._reset_bookkeeping(only_episode_counters)
        return self.observation()

    def _resolved_action(self, action, r):
        """(path, initial_slot) as RMSAEnv.step sees them"""
        return int(action[0]), int(action[1])

    def _device_step(self, action):
        path, initial_slot = int(action[0]), int(action[1])
        return self._batched.run("external", 1, actions=np.array([[path, initial_slot]], np.int32),
                                 outputs=("accepted", "done", "reward", "network_compactness",
                                          "network_compactness_difference", "avg_link_compactness",
                                          "avg_link_utilization"))

    def step(self, action):
        """``rmsa_env.py:222-341`` -> (observation, reward, done, info)"""
        if self._batched.batch_size != 1:
            raise RuntimeError("step() needs a batch-1 environment: use the batched API for B > 1")
        served = self.current_service
        candidates = self._candidates
        r = self._device_step(action)
        served.accepted = bool(r["accepted"][0, 0])
        # rmsa_env.py:226, 261-267, 271, 509: every action counted, the accepted ones again -- and their slots twice (step and
        # _provision_path both add one)
        path, initial_slot = self._resolved_action(action, r)
        self.actions_output[path, initial_slot] += 1
        if served.accepted:
            self.actions_taken[path, initial_slot] += 1
            if self.bit_rate_selection == "discrete":
                self.slots_provisioned_histogram[self.get_number_slots(candidates[path])] += 2
        else:
            self.actions_taken[self.k_paths, self.num_spectrum_resources] += 1
        info = self._info(float(r["network_compactness"][0, 0]), float(r["network_compactness_difference"][0, 0]),
                          float(r["avg_link_compactness"][0, 0]), float(r["avg_link_utilization"][0, 0]))
        reward = r["reward"][0, 0]
        reward = int(reward) if float(reward).is_integer() else float(reward)
        self._sync()
        self._count_request()
        self._last_served = served
        return self.observation(), reward, bool(r["done"][0, 0]), info

    def _info(self, compactness, compactness_difference, avg_link_compactness, avg_link_utilization):
        """The info dict of ``rmsa_env.py:293-332``: integer ratios from the device counters (Python int
        division, as in the reference), floats from the device statistics."""
        b, i = self._batched, self._index
        c = {k: int(v[i]) for k, v in b.counters().items()}
        # the reference builds info BEFORE _next_service() (rmsa_env.py:293-335); the device step already
        # generated the next request, so take its contribution out of the request-side counters again
        nxt = b.requests()[i]
        nxt_rate = int(nxt["bit_rate"])
        c["services_processed"] -= 1
        c["episode_services_processed"] -= 1
        c["bit_rate_requested"] -= nxt_rate
        c["episode_bit_rate_requested"] -= nxt_rate
        info = {
            "service_blocking_rate": (c["services_processed"] - c["services_accepted"]) / c["services_processed"],
            "episode_service_blocking_rate": (c["episode_services_processed"] - c["episode_services_accepted"])
            / c["episode_services_processed"],
            "bit_rate_blocking_rate": (c["bit_rate_requested"] - c["bit_rate_provisioned"]) / c["bit_rate_requested"],
            "episode_bit_rate_blocking_rate": (c["episode_bit_rate_requested"] - c["episode_bit_rate_provisioned"])
            / c["episode_bit_rate_requested"],
            "network_compactness": compactness,
            "network_compactness_difference": compactness_difference,
        }
        # np.mean over the links, evaluated on the device at the reference's point in the step
        info["avg_link_compactness"] = avg_link_compactness
        info["avg_link_utilization"] = avg_link_utilization
        if self.bit_rate_selection != "discrete":   # rmsa_env.py:276, 327: the per-rate keys exist in discrete mode only
            return info
        h = b.bit_rate_hist()
        blocking = {}
        for k, rate in enumerate(self.bit_rates):
            req, prov = int(h["requested"][i, k]) - (1 if rate == nxt_rate else 0), int(h["provisioned"][i, k])
            blocking[rate] = (req - prov) / req if req > 0 else 0.0
        for rate, v in blocking.items():
            info[f"bit_rate_blocking_{rate}"] = v
        info["fairness"] = max(blocking.values()) - min(blocking.values())
        return info

    def close(self):
        if self._batched.batch_size == 1:
            self._batched.close()


class DeepRMSAEnv(RMSAEnv):
    """Drop-in for ``DeepRMSAEnv`` (``deeprmsa_env.py:9``)."""

    _batched_cls = BatchedDeepRMSAEnv

    def __init__(self, topology=None, j: int = 1, episode_length: int = 1000, mean_service_holding_time: float = 25.0,
                 mean_service_inter_arrival_time: float = 0.1, num_spectrum_resources: int = 100,
                 node_request_probabilities=None, seed=None, allow_rejection: bool = False, device: int = 0,
                 _batched=None, _index: int = 0):
        if _batched is None:
            _batched = BatchedDeepRMSAEnv(topology, 1, j=j, episode_length=episode_length,
                                          mean_service_holding_time=mean_service_holding_time,
                                          mean_service_inter_arrival_time=mean_service_inter_arrival_time,
                                          num_spectrum_resources=num_spectrum_resources,
                                          node_request_probabilities=node_request_probabilities, seed=seed,
                                          allow_rejection=allow_rejection, device=device)
        self._init_view(_batched, _index)
        shape = 1 + 2 * self._ft.num_nodes + (2 * self.j + 3) * self.k_paths
        self.observation_space = _box(-2 ** 30, 2 ** 30, (shape,), np.float64)
        self.action_space = _discrete(self.k_paths * self.j + self.reject_action)
        self.action_space.seed(self.rand_seed)

    def observation(self):
        """``deeprmsa_env.py:60-121`` (built by ``orlg_deeprmsa_obs_kernel``)."""
        return self._batched.observation()[self._index]

    def reward(self):
        return 1 if self.current_service.accepted else -1

    def _device_step(self, action):
        return self._batched.run("deeprmsa_external", 1, actions=np.array([int(action)], np.int32),
                                 outputs=("accepted", "done", "reward", "network_compactness",
                                          "network_compactness_difference", "avg_link_compactness",
                                          "avg_link_utilization", "act_path", "act_slot"))

    def _resolved_action(self, action, r):
        """``deeprmsa_env.py:48-58``: the (route, first slot of the block) the action stands for, (k, S) for a rejection"""
        return int(r["act_path"][0, 0]), int(r["act_slot"][0, 0])

    def _get_route_block_id(self, action: int) -> Tuple[int, int]:
        return action // self.j, action % self.j


class SimpleMatrixObservation:
    """``rmsa_env.py:940-971``: observation = one-hot endpoints + the E x S free-slot matrix (built on the device)."""

    def __init__(self, env: RMSAEnv):
        self.env = env
        shape = env.topology.number_of_nodes() * 2 + env.topology.number_of_edges() * env.num_spectrum_resources
        self.observation_space = _box(0, 1, (shape,), np.uint8)
        self.action_space = env.action_space

    def __getattr__(self, name):
        return getattr(self.env, name)

    def observation(self, observation=None):
        return self.env._batched.simple_matrix_observation()[self.env._index]

    def reset(self, **kw):
        self.env.reset(**kw)
        return self.observation()

    def step(self, action):
        _, reward, done, info = self.env.step(action)
        return self.observation(), reward, done, info


class PathOnlyFirstFitAction:
    """``rmsa_env.py:974-1008``: the agent chooses the path, the slot is the first fit (resolved on the device)."""

    def __init__(self, env):
        self.env = env
        inner = env
        while not isinstance(inner, RMSAEnv):
            inner = inner.env
        self._inner = inner
        self.action_space = _discrete(inner.k_paths + inner.reject_action)
        self.observation_space = getattr(env, "observation_space", None)

    def __getattr__(self, name):
        return getattr(self.env, name)

    def action(self, action) -> Tuple[int, int]:
        e = self._inner
        if action < e.k_paths:
            path = e.k_shortest_paths[e.current_service.source, e.current_service.destination][action]
            n = e.get_number_slots(path)
            for s in range(0, e.topology.graph["num_spectrum_resources"] - n):
                if e.is_path_free(path, s, n):
                    return (action, s)
        return (e.topology.graph["k_paths"], e.topology.graph["num_spectrum_resources"])

    def step(self, action):
        return self.env.step(self.action(action))

    def reset(self, **kw):
        return self.env.reset(**kw)


# --------------------------------------------------------------------------------------- heuristics
def shortest_path_first_fit(env: RMSAEnv) -> Tuple[int, int]:
    """SP-FF (``rmsa_env.py:854-871``): first fit on the shortest path only; note the exclusive bound."""
    S, k = env.topology.graph["num_spectrum_resources"], env.topology.graph["k_paths"]
    path = env.k_shortest_paths[env.current_service.source, env.current_service.destination][0]
    n = env.get_number_slots(path)
    for s in range(0, S - n):
        if env.is_path_free(path, s, n):
            return (0, s)
    return (k, S)


def shortest_available_path_first_fit(env: RMSAEnv) -> Tuple[int, int]:
    """SAP-FF (``rmsa_env.py:901-913``)."""
    S, k = env.topology.graph["num_spectrum_resources"], env.topology.graph["k_paths"]
    for idp, path in enumerate(env.k_shortest_paths[env.current_service.source, env.current_service.destination]):
        n = env.get_number_slots(path)
        for s in range(0, S - n):
            if env.is_path_free(path, s, n):
                return (idp, s)
    return (k, S)


def least_loaded_path_first_fit(env: RMSAEnv) -> Tuple[int, int]:
    """LLP-FF (``rmsa_env.py:916-937``): among paths with a first fit, the one with most free slots (strict >)."""
    S, k = env.topology.graph["num_spectrum_resources"], env.topology.graph["k_paths"]
    best, action = 0, (k, S)
    for idp, path in enumerate(env.k_shortest_paths[env.current_service.source, env.current_service.destination]):
        n = env.get_number_slots(path)
        for s in range(0, S - n):
            if env.is_path_free(path, s, n):
                free = int(np.sum(env.get_available_slots(path)))
                if free > best:
                    best, action = free, (idp, s)
                break
    return action


def deeprmsa_shortest_path_first_fit(env: DeepRMSAEnv) -> int:
    """``deeprmsa_env.py:135-143``"""
    if not env.allow_rejection:
        return 0
    starts, _ = env.get_available_blocks(0)
    return 0 if len(starts) > 0 else env.k_paths * env.j


def deeprmsa_shortest_available_path_first_fit(env: DeepRMSAEnv) -> int:
    """``deeprmsa_env.py:146-155``"""
    for idp, _ in enumerate(env.k_shortest_paths[env.current_service.source, env.current_service.destination]):
        starts, _ = env.get_available_blocks(idp)
        if len(starts) > 0:
            return idp * env.j
    return env.k_paths * env.j


def random_policy(env):
    return env.action_space.sample()


def evaluate_heuristic(env, heuristic, n_eval_episodes=10, render=False, callback=None, reward_threshold=None,
                       return_episode_rewards=False):
    """``utils.py:124-162``, accepting both 4-tuple (RMSA / DeepRMSA) and 5-tuple (PhyRMSA) steps -- the
    reference's own loop unpacks five values and therefore cannot drive RMSAEnv."""
    episode_rewards, episode_lengths = [], []
    for _ in range(n_eval_episodes):
        env.reset()
        done, episode_reward, episode_length = False, 0.0, 0
        while not done:
            out = env.step(heuristic(env))
            reward, done = out[1], out[2]
            episode_reward += reward
            if callback is not None:
                callback(locals(), globals())
            episode_length += 1
            if render:
                env.render()
        episode_rewards.append(episode_reward)
        episode_lengths.append(episode_length)
    mean_reward, std_reward = np.mean(episode_rewards), np.std(episode_rewards)
    if reward_threshold is not None:
        assert mean_reward > reward_threshold, "Mean reward below threshold: {:.2f} < {:.2f}".format(
            mean_reward, reward_threshold)
    if return_episode_rewards:
        return episode_rewards, episode_lengths
    return mean_reward, std_reward
