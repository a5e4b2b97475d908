#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [--only topologies|rmsa|deeprmsa|phy|osnr]

The reference (pure Python) is imported from /root/reference with a tiny in-memory
``gym`` stand-in (gym itself is not installed here; the reference only needs the base
classes, four space types with ``.seed()`` and ``register``).  Nothing from the
reference is copied: the fixtures are *data* -- frozen topologies (node / link / k-path
tables read out of the shipped pickles) and per-step input/output vectors.

Outputs
  topologies/<name>.json      frozen topology tables (see optical_rl_gym_amd.topology)
  rmsa_*.npz                  RMSAEnv per-step traces (requests, actions, counters, info floats)
  deeprmsa_*.npz              DeepRMSAEnv traces incl. the observation vector
  phy_*.npz, tables/*.npz     PhyRMSAEnv traces and the QoT tables (uint8 / float64)
  osnr_grid.npz               GN-model inputs / expected GSNR
"""
import argparse
import json
import os
import sys
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# --------------------------------------------------------------------------- gym stand-in
def install_gym_stub():
    gym = types.ModuleType("gym")

    class Env:
        pass

    class Wrapper:
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.env, name)

    class ObservationWrapper(Wrapper):
        pass

    class ActionWrapper(Wrapper):
        pass

    class RewardWrapper(Wrapper):
        pass

    gym.Env, gym.Wrapper = Env, Wrapper
    gym.ObservationWrapper, gym.ActionWrapper, gym.RewardWrapper = (
        ObservationWrapper,
        ActionWrapper,
        RewardWrapper,
    )
    spaces = types.ModuleType("gym.spaces")

    class _Space:
        def __init__(self, *a, **k):
            self.args, self.kw = a, k
            self.shape = k.get("shape")

        def seed(self, s=None):
            pass

    for n in ("MultiDiscrete", "Discrete", "Dict", "Box"):
        setattr(spaces, n, type(n, (_Space,), {}))
    gym.spaces = spaces
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda **k: None
    envs = types.ModuleType("gym.envs")
    envs.registration = reg
    gym.envs = envs
    sys.modules.update(
        {"gym": gym, "gym.spaces": spaces, "gym.envs": envs, "gym.envs.registration": reg}
    )
    if REF not in sys.path:
        sys.path.insert(0, REF)


def load_pickled_topology(spec):
    """The topology graph the reference's environments take (``topology.graph["ksp" | "modulations" | ...]``).

    The shipped ``examples/topologies/*.h5`` files are pickles: they are NOT loaded (unpickling public files can run
    arbitrary code).  The graph is rebuilt from the link-list text file next to them with the reference's own
    generator, ``examples/create_topology.py::get_topology`` (imported, unmodified) and its modulation table --
    the same call that produced the pickles (``tests/test_topology_io.py`` holds our own front-end to the same
    frozen tables).  The historic name of this helper is kept for the call sites below."""
    import contextlib
    import io
    txt, k = spec
    ex = os.path.join(REF, "examples")
    if ex not in sys.path:
        sys.path.insert(0, ex)
    import create_topology as ct   # main-guarded script: importing it only defines get_topology + the modulation table
    name = os.path.splitext(txt)[0].upper()   # the shipped pickles carry the upper-case name (e.g. "NSFNET_CHEN")
    with contextlib.redirect_stdout(io.StringIO()):   # get_topology prints every path
        return ct.get_topology(os.path.join(ex, "topologies", txt), name, ct.modulations, k)


# --------------------------------------------------------------------------- topologies
TOPOLOGIES = {   # fixture name -> (link-list file, k shortest paths)
    "nsfnet_chen_5-paths_6-modulations": ("nsfnet_chen.txt", 5),
    "us14_3-paths_6-modulations": ("us14.txt", 3),
    "jpn12_3-paths_6-modulations": ("jpn12.txt", 3),
    "jpn12_5-paths_6-modulations": ("jpn12.txt", 5),
    "spn_3-paths_6-modulations": ("spn.txt", 3),
}


def freeze_topology(topo):
    nodes = [str(n) for n in topo.nodes()]
    edges = []
    for a, b in topo.edges():
        d = topo[a][b]
        edges.append([str(a), str(b), int(d["index"]), int(d["id"]), float(d["length"])])
    paths = {}
    for i, a in enumerate(nodes):
        for j, b in enumerate(nodes):
            if i < j:
                lst = topo.graph["ksp"][a, b]
                assert lst is topo.graph["ksp"][b, a]
                paths[f"{a},{b}"] = [
                    [
                        int(p.path_id),
                        int(p.hops),
                        float(p.length),
                        int(p.best_modulation.spectral_efficiency),
                        [str(n) for n in p.node_list],
                    ]
                    for p in lst
                ]
    return {
        "name": topo.graph["name"],
        "nodes": nodes,
        "node_indices": [str(n) for n in topo.graph["node_indices"]],
        "k_paths": int(topo.graph["k_paths"]),
        "edges": edges,
        "modulations": [
            [m.name, float(m.maximum_length), int(m.spectral_efficiency)]
            for m in topo.graph["modulations"]
        ],
        "paths": paths,
    }


def gen_topologies():
    os.makedirs(os.path.join(HERE, "topologies"), exist_ok=True)
    for name, fname in TOPOLOGIES.items():
        fz = freeze_topology(load_pickled_topology(fname))
        with open(os.path.join(HERE, "topologies", name + ".json"), "w") as f:
            json.dump(fz, f, separators=(",", ":"))
        print("topology", name, len(fz["nodes"]), "nodes", len(fz["edges"]), "links")


# --------------------------------------------------------------------------- RMSA traces
def occ_crc(avail):
    """crc32 of the little-endian bit-packed (link-major) free-slot matrix."""
    return zlib.crc32(np.packbits(avail.astype(np.uint8), axis=1, bitorder="little").tobytes())


class Recorder:
    def __init__(self):
        self.cols = {}

    def add(self, **kw):
        for k, v in kw.items():
            self.cols.setdefault(k, []).append(v)

    def arrays(self):
        out = {}
        for k, v in self.cols.items():
            a = np.asarray(v)
            if a.dtype == np.int64 and k in ("src_id", "dst_id", "act_path", "act_slot", "bit_rate", "service_id"):
                a = a.astype(np.int32)
            if a.dtype == bool:
                a = a.astype(np.uint8)
            out[k] = a
        return out


def run_rmsa_trace(topo, env_kwargs, policy, n_steps, reset_on_done, rec_links=True, random_actions_seed=None):
    from optical_rl_gym.envs import rmsa_env as R

    env = R.RMSAEnv(topology=topo, **env_kwargs)
    pol = {
        "sp_ff": R.shortest_path_first_fit,
        "sap_ff": R.shortest_available_path_first_fit,
        "llp_ff": R.least_loaded_path_first_fit,
    }.get(policy)
    arng = np.random.default_rng(random_actions_seed) if policy == "random" else None
    rec = Recorder()
    k, S = env.k_paths, env.num_spectrum_resources
    for _ in range(n_steps):
        s = env.current_service
        if pol is not None:
            a = pol(env)
        else:
            # mix of in-range and out-of-range (= rejection) actions
            a = (int(arng.integers(0, k + 1)), int(arng.integers(0, S + 1)))
        _, reward, done, info = env.step(a)
        av = env.topology.graph["available_slots"]
        rec.add(
            service_id=s.service_id, src_id=s.source_id, dst_id=s.destination_id, bit_rate=s.bit_rate,
            arrival=s.arrival_time, holding=s.holding_time,
            act_path=a[0], act_slot=a[1], accepted=bool(s.accepted), reward=float(reward), done=bool(done),
            services_processed=env.services_processed, services_accepted=env.services_accepted,
            episode_services_processed=env.episode_services_processed,
            episode_services_accepted=env.episode_services_accepted,
            bit_rate_requested=env.bit_rate_requested, bit_rate_provisioned=env.bit_rate_provisioned,
            episode_bit_rate_requested=env.episode_bit_rate_requested,
            episode_bit_rate_provisioned=env.episode_bit_rate_provisioned,
            network_compactness=float(info["network_compactness"]),
            network_compactness_difference=float(info["network_compactness_difference"]),
            avg_link_compactness=float(info["avg_link_compactness"]),
            avg_link_utilization=float(info["avg_link_utilization"]),
            fairness=float(info.get("fairness", 0.0)),   # (discrete mode only: rmsa_env.py:327-332)
            free_total=int(av.sum()), occ_crc=occ_crc(av), current_time=env.current_time,
            graph_throughput=float(env.topology.graph["throughput"]),
            graph_compactness=float(env.topology.graph["compactness"]),
        )
        if done and reset_on_done:
            env.reset()
    out = rec.arrays()
    edges = sorted(env.topology.edges(), key=lambda e: env.topology[e[0]][e[1]]["index"])
    out["final_link_utilization"] = np.array([env.topology[a][b]["utilization"] for a, b in edges])
    out["final_link_external_fragmentation"] = np.array(
        [env.topology[a][b]["external_fragmentation"] for a, b in edges])
    out["final_link_compactness"] = np.array([env.topology[a][b]["compactness"] for a, b in edges])
    out["final_link_last_update"] = np.array([env.topology[a][b]["last_update"] for a, b in edges])
    out["final_available_slots"] = np.packbits(
        env.topology.graph["available_slots"].astype(np.uint8), axis=1, bitorder="little")
    if env.bit_rate_selection == "discrete":   # the histograms and the per-rate info keys exist in this mode only
        out["final_bit_rate_requested_hist"] = np.array(
            [env.bit_rate_requested_histogram[b] for b in env.bit_rates], dtype=np.int64)
        out["final_bit_rate_provisioned_hist"] = np.array(
            [env.bit_rate_provisioned_histogram[b] for b in env.bit_rates], dtype=np.int64)
        out["final_episode_bit_rate_requested_hist"] = np.array(
            [env.episode_bit_rate_requested_histogram[b] for b in env.bit_rates], dtype=np.int64)
        out["final_episode_bit_rate_provisioned_hist"] = np.array(
            [env.episode_bit_rate_provisioned_histogram[b] for b in env.bit_rates], dtype=np.int64)
        # last step's per-bit-rate blocking (info keys bit_rate_blocking_<rate>)
        out["final_bit_rate_blocking"] = np.array([info[f"bit_rate_blocking_{b}"] for b in env.bit_rates])
    # the request pending after the last step (what the next heuristic call would see)
    s = env.current_service
    out["pending"] = np.array([s.source_id, s.destination_id, s.bit_rate, s.service_id], dtype=np.int64)
    out["pending_times"] = np.array([s.arrival_time, s.holding_time])
    return out


RMSA_BASE = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25,
                 episode_length=1000, allow_rejection=False)

DEEPRMSA_NODE_PROBS = [0.01801802, 0.04004004, 0.05305305, 0.01901902, 0.04504505, 0.02402402, 0.06706707,
                       0.08908909, 0.13813814, 0.12212212, 0.07607608, 0.12012012, 0.01901902, 0.16916917]

RMSA_CASES = [
    # name, topology, env kwargs override, policy, steps, reset_on_done
    ("rmsa_nsfnet_s10_sapff", "nsfnet_chen_5-paths_6-modulations", dict(seed=10), "sap_ff", 3000, False),
    ("rmsa_nsfnet_s11_sapff", "nsfnet_chen_5-paths_6-modulations", dict(seed=11), "sap_ff", 1500, False),
    ("rmsa_nsfnet_s10_spff", "nsfnet_chen_5-paths_6-modulations", dict(seed=10), "sp_ff", 1500, False),
    ("rmsa_nsfnet_s12_llpff", "nsfnet_chen_5-paths_6-modulations", dict(seed=12), "llp_ff", 1500, False),
    ("rmsa_nsfnet_s10_sapff_reset", "nsfnet_chen_5-paths_6-modulations",
     dict(seed=10, episode_length=100), "sap_ff", 1200, True),
    ("rmsa_nsfnet_s13_random", "nsfnet_chen_5-paths_6-modulations",
     dict(seed=13, allow_rejection=True, load=20), "random", 1500, False),
    ("rmsa_nsfnet_s7_sapff_nodeprobs", "nsfnet_chen_5-paths_6-modulations",
     dict(seed=7, node_request_probabilities=np.array(DEEPRMSA_NODE_PROBS), load=80,
          bit_rates=[25, 50, 75, 100], bit_rate_probabilities=[0.4, 0.3, 0.2, 0.1],
          num_spectrum_resources=100, episode_length=50), "sap_ff", 1000, True),
    ("rmsa_jpn12k5_s3_sapff", "jpn12_5-paths_6-modulations", dict(seed=3, load=120), "sap_ff", 1000, False),
    ("rmsa_nsfnet_s10_sapff_continuous", "nsfnet_chen_5-paths_6-modulations",
     dict(seed=10, load=300, num_spectrum_resources=100, bit_rate_selection="continuous", bit_rate_lower_bound=25,
          bit_rate_higher_bound=100, episode_length=200), "sap_ff", 1500, True),
    ("rmsa_us14_s4_llpff_continuous", "us14_3-paths_6-modulations",
     dict(seed=4, load=150, num_spectrum_resources=128, bit_rate_selection="continuous", bit_rate_lower_bound=100,
          bit_rate_higher_bound=300), "llp_ff", 800, False),
    ("rmsa_us14_s5_llpff", "us14_3-paths_6-modulations", dict(seed=5, load=60, num_spectrum_resources=192),
     "llp_ff", 1000, False),
    ("rmsa_spn_s2_sapff", "spn_3-paths_6-modulations", dict(seed=2, load=400, num_spectrum_resources=64),
     "sap_ff", 800, False),
]


def _jsonable(d):
    out = {}
    for k, v in d.items():
        out[k] = v.tolist() if isinstance(v, np.ndarray) else v
    return out


def gen_rmsa():
    for name, tname, over, policy, steps, reset in RMSA_CASES:
        kw = dict(RMSA_BASE)
        kw.update(over)
        topo = load_pickled_topology(TOPOLOGIES[tname])
        out = run_rmsa_trace(topo, kw, policy, steps, reset,
                             random_actions_seed=kw.get("seed", 0) + 1000)
        meta = dict(topology=tname, env_kwargs=_jsonable(kw), policy=policy, steps=steps, reset_on_done=reset,
                    random_actions_seed=kw.get("seed", 0) + 1000)
        out["meta"] = np.array(json.dumps(meta))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "accepted", int(out["services_accepted"][-1]), "/", int(out["services_processed"][-1]),
              "compactness", out["network_compactness"][-1])


# --------------------------------------------------------------------------- seed() in the middle of a run
def gen_seed():
    """OpticalNetworkEnv.seed (optical_network_env.py:266-271) called between two steps: a fresh random.Random(seed), nothing
    else changes -- the pending request stays, the next arrival is the new generator's first draw.  RMSAEnv and PhyRMSAEnv."""
    from optical_rl_gym.envs import rmsa_env as R
    topo = load_pickled_topology(TOPOLOGIES["nsfnet_chen_5-paths_6-modulations"])
    kw = dict(RMSA_BASE, seed=10)
    env = R.RMSAEnv(topology=topo, **kw)
    rows = []
    for t in range(120):
        if t == 40:
            env.seed(77)
        if t == 80:
            env.seed()          # the default: 41
        s = env.current_service
        a = R.shortest_available_path_first_fit(env)
        env.step(a)
        rows.append((s.source_id, s.destination_id, s.bit_rate, s.arrival_time, s.holding_time, a[0], a[1], int(s.accepted)))
    r = np.array(rows, dtype=np.float64)
    meta = dict(topology="nsfnet_chen_5-paths_6-modulations", env_kwargs=_jsonable(kw), policy="sap_ff", steps=120,
                reseed={"40": 77, "80": None})
    np.savez_compressed(os.path.join(HERE, "seed_rmsa_nsfnet_s10.npz"), src=r[:, 0].astype(np.int32), dst=r[:, 1].astype(np.int32),
                        bit_rate=r[:, 2].astype(np.int32), arrival=r[:, 3], holding=r[:, 4], act_path=r[:, 5].astype(np.int32),
                        act_slot=r[:, 6].astype(np.int32), accepted=r[:, 7].astype(np.uint8), meta=np.array(json.dumps(meta)))
    print("seed_rmsa_nsfnet_s10: accepted", int(r[:, 7].sum()), "of 120; arrival[41] =", r[41, 3])


def _hist_rows(d):
    """defaultdict(int) -> sorted (key, count) rows"""
    return np.array(sorted((int(k), int(v)) for k, v in d.items()), dtype=np.int64).reshape(-1, 2)


def gen_bookkeeping():
    """The arrays the reference keeps beside the simulation and no info key reads: actions_output / actions_taken
    (rmsa_env.py:185-196, 226, 261, 271; the episode_ twins are only ever re-zeroed: :348-360), slots_requested_histogram
    (:685-686, 387) and slots_provisioned_histogram with its double increment (:265-267 and 509), and PhyRMSAEnv's BVT counters
    (phy_rmsa_env.py:153-156, 603-608).  One RMSA run with random in- and out-of-range actions across episode resets and one
    PhyRMSA run: the action traces and the arrays after the last step."""
    from optical_rl_gym.envs import rmsa_env as R
    from optical_rl_gym.envs import phy_rmsa_env as P
    out = {}
    topo = load_pickled_topology(TOPOLOGIES["nsfnet_chen_5-paths_6-modulations"])
    kw = dict(RMSA_BASE, seed=13, allow_rejection=True, load=40, episode_length=250)
    env = R.RMSAEnv(topology=topo, **kw)
    arng = np.random.default_rng(5)
    k, S = env.k_paths, env.num_spectrum_resources
    acts, acc = [], []
    for t in range(900):
        if t % 3 == 0:
            a = (int(arng.integers(0, k + 1)), int(arng.integers(0, S + 1)))
        else:
            a = R.shortest_available_path_first_fit(env)
        s = env.current_service
        _, _, done, _ = env.step(a)
        acts.append((int(a[0]), int(a[1]))); acc.append(bool(s.accepted))
        if done:
            env.reset()
    out["rmsa_actions"] = np.array(acts, np.int32)
    out["rmsa_accepted"] = np.array(acc, np.uint8)
    for name in ("actions_output", "actions_taken", "episode_actions_output", "episode_actions_taken"):
        out["rmsa_" + name] = np.asarray(getattr(env, name), dtype=np.int64)
    for name in ("slots_requested_histogram", "episode_slots_requested_histogram", "slots_provisioned_histogram",
                 "episode_slots_provisioned_histogram"):
        out["rmsa_" + name] = _hist_rows(getattr(env, name))
    meta = dict(rmsa=dict(topology="nsfnet_chen_5-paths_6-modulations", env_kwargs=_jsonable(kw), steps=900))

    tab = "us14_k3"
    pkw = dict(PHY_BASE, seed=21, load=1400)
    ptopo = load_pickled_topology(TOPOLOGIES[PHY_TABLES[tab][2]])
    conn, mod, gsnr = load_phy_tables(tab)
    penv = P.PhyRMSAEnv(topology=ptopo, modulation_level=mod, connections_detail=conn, gsnr=gsnr, **pkw)
    MAXCH = 12
    paths, chans, used, free, cap, pacc = [], [], [], [], [], []
    for t in range(500):
        a = P.phy_aware_bmff_rmsa(penv)     # (bmff consults the virtual layer: physical and virtual acceptances mix)
        chosen = [tuple(c) for c in a[1]]
        s = penv.current_service
        _, _, done, _, _ = penv.step(a)
        row_c, row_u, row_f, row_k = [-1] * MAXCH, [0.0] * MAXCH, [0.0] * MAXCH, [0] * MAXCH
        for i, c in enumerate(chosen):
            row_c[i], row_u[i], row_f[i], row_k[i] = int(c[0]), float(c[1]), float(c[2]), int(c[3])
        paths.append(int(a[0])); chans.append(row_c); used.append(row_u); free.append(row_f); cap.append(row_k)
        pacc.append(bool(s.accepted))
        if done:
            penv.reset()
    out["phy_act_path"] = np.array(paths, np.int32)
    out["phy_channels"] = np.array(chans, np.int16)
    out["phy_ch_used"] = np.array(used); out["phy_ch_free"] = np.array(free); out["phy_ch_cap"] = np.array(cap, np.int16)
    out["phy_accepted"] = np.array(pacc, np.uint8)
    out["phy_bvts"] = np.asarray(penv.bvts, dtype=np.int64)
    meta["phy"] = dict(topology=PHY_TABLES[tab][2], tables=tab, env_kwargs=_jsonable(pkw), policy="bmff", steps=500)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "bookkeeping.npz"), **out)
    print("bookkeeping: rmsa accepted", int(out["rmsa_accepted"].sum()), "of 900, actions_taken sum", int(out["rmsa_actions_taken"].sum()),
          "slots_provisioned", out["rmsa_slots_provisioned_histogram"].tolist(), "| phy accepted", int(out["phy_accepted"].sum()),
          "bvts", out["phy_bvts"].sum(axis=(1, 2)).tolist())


# --------------------------------------------------------------------------- RMSA wrappers
def gen_wrappers():
    """SimpleMatrixObservation (rmsa_env.py:940-971) and PathOnlyFirstFitAction (:974-1008) on RMSA-v0."""
    from optical_rl_gym.envs import rmsa_env as R
    topo = load_pickled_topology(TOPOLOGIES["nsfnet_chen_5-paths_6-modulations"])
    kw = dict(RMSA_BASE, seed=21, num_spectrum_resources=128, load=40)
    env = R.PathOnlyFirstFitAction(R.SimpleMatrixObservation(R.RMSAEnv(topology=topo, **kw)))
    inner = env.env.env
    arng = np.random.default_rng(77)
    rec, obs = Recorder(), []
    for _ in range(600):
        a = int(arng.integers(0, inner.k_paths + 1))
        s = inner.current_service
        resolved = env.action(a)
        _, reward, done, info = env.step(a)
        # the gym stand-in's ObservationWrapper does not intercept step(): call the wrapper's observation() directly
        o = env.env.observation(None)
        obs.append(np.asarray(o, dtype=np.float64))
        rec.add(action=a, act_path=int(resolved[0]), act_slot=int(resolved[1]), accepted=bool(s.accepted),
                reward=float(reward), services_accepted=inner.services_accepted)
    out = rec.arrays()
    out["obs"] = np.stack(obs)[:, :].astype(np.uint8)  # values are 0/1
    out["meta"] = np.array(json.dumps(dict(topology="nsfnet_chen_5-paths_6-modulations", env_kwargs=_jsonable(kw), steps=600)))
    np.savez_compressed(os.path.join(HERE, "wrappers_nsfnet_s21.npz"), **out)
    print("wrappers: accepted", int(out["services_accepted"][-1]), "obs dim", out["obs"].shape[1])


# --------------------------------------------------------------------------- DeepRMSA traces
def run_deeprmsa_trace(topo, env_kwargs, policy, n_steps, reset_on_done, actions_seed):
    from optical_rl_gym.envs import deeprmsa_env as D

    env = D.DeepRMSAEnv(topology=topo, **env_kwargs)
    pol = {"deeprmsa_sp_ff": D.shortest_path_first_fit, "deeprmsa_sap_ff": D.shortest_available_path_first_fit}.get(policy)
    arng = np.random.default_rng(actions_seed)
    rec = Recorder()
    obs0 = env.observation()
    obs = []
    for _ in range(n_steps):
        s = env.current_service
        if pol is not None:
            a = int(pol(env))
        else:
            a = int(arng.integers(0, env.k_paths * env.j + 1))  # k*j = rejection
        o, reward, done, info = env.step(a)
        obs.append(np.asarray(o, dtype=np.float64))
        av = env.topology.graph["available_slots"]
        rec.add(
            service_id=s.service_id, src_id=s.source_id, dst_id=s.destination_id, bit_rate=s.bit_rate,
            arrival=s.arrival_time, holding=s.holding_time, action=a, accepted=bool(s.accepted),
            act_slot=int(s.initial_slot) if s.accepted else env.num_spectrum_resources,
            reward=float(reward), done=bool(done),
            services_processed=env.services_processed, services_accepted=env.services_accepted,
            episode_services_processed=env.episode_services_processed,
            episode_services_accepted=env.episode_services_accepted,
            bit_rate_requested=env.bit_rate_requested, bit_rate_provisioned=env.bit_rate_provisioned,
            network_compactness=float(info["network_compactness"]),
            free_total=int(av.sum()), occ_crc=occ_crc(av), current_time=env.current_time,
        )
        if done and reset_on_done:
            env.reset()
    out = rec.arrays()
    out["action"] = out["action"].astype(np.int32)
    out["obs0"] = np.asarray(obs0, dtype=np.float64)
    out["obs"] = np.stack(obs)
    out["final_available_slots"] = np.packbits(
        env.topology.graph["available_slots"].astype(np.uint8), axis=1, bitorder="little")
    return out


DEEPRMSA_BASE = dict(seed=10, allow_rejection=False, mean_service_holding_time=7.5,
                     mean_service_inter_arrival_time=1.0 / 12.0, j=1, episode_length=50,
                     node_request_probabilities=np.array(DEEPRMSA_NODE_PROBS))

DEEPRMSA_CASES = [
    # tests/test_deeprmsa.py:30-47 configuration (S = the class default 100) and the BASELINE S=320 variant
    ("deeprmsa_nsfnet_s10_sapff", "nsfnet_chen_5-paths_6-modulations", dict(), "deeprmsa_sap_ff", 1000, True),
    ("deeprmsa_nsfnet_s10_spff", "nsfnet_chen_5-paths_6-modulations", dict(), "deeprmsa_sp_ff", 600, True),
    ("deeprmsa_nsfnet_s10_sapff_320", "nsfnet_chen_5-paths_6-modulations",
     dict(num_spectrum_resources=320, node_request_probabilities=None, mean_service_inter_arrival_time=1.0 / 24.0),
     "deeprmsa_sap_ff", 1000, True),
    # SURVEY 8(d) config 4 / BASELINE configs[3] exactly: S = 320, inter-arrival 1/12, the DeepRMSA node probabilities
    ("deeprmsa_nsfnet_s10_sapff_320_config4", "nsfnet_chen_5-paths_6-modulations", dict(num_spectrum_resources=320),
     "deeprmsa_sap_ff", 1000, True),
    ("deeprmsa_nsfnet_s4_random_j3", "nsfnet_chen_5-paths_6-modulations",
     dict(seed=4, j=3, num_spectrum_resources=160, mean_service_inter_arrival_time=1.0 / 10.0), "random", 1000, True),
    ("deeprmsa_jpn12_s6_random_j2", "jpn12_3-paths_6-modulations",
     dict(seed=6, j=2, num_spectrum_resources=100, node_request_probabilities=None,
          mean_service_inter_arrival_time=1.0 / 20.0, episode_length=200), "random", 800, True),
]


def gen_deeprmsa():
    for name, tname, over, policy, steps, reset in DEEPRMSA_CASES:
        kw = dict(DEEPRMSA_BASE)
        kw.update(over)
        topo = load_pickled_topology(TOPOLOGIES[tname])
        out = run_deeprmsa_trace(topo, kw, policy, steps, reset, kw.get("seed", 0) + 2000)
        meta = dict(topology=tname, env_kwargs=_jsonable(kw), policy=policy, steps=steps, reset_on_done=reset)
        out["meta"] = np.array(json.dumps(meta))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "accepted", int(out["services_accepted"][-1]), "/", int(out["services_processed"][-1]),
              "obs dim", out["obs"].shape[1])


# --------------------------------------------------------------------------- PhyRMSA (QoT-aware) traces
PHY_TABLES = {
    # name: (mat file, struct name, topology fixture)
    "us14_k3": ("Results_K3SP_FRP_SLC_CBG_USB14.mat", "Results_K3SP_FRP_SLC_CBG_USB14", "us14_3-paths_6-modulations"),
    "jpn12_k3": ("Results_K3SP_FRP_SLC_CBG_JPN12.mat", "Results_K3SP_FRP_SLC_CBG_JPN12", "jpn12_3-paths_6-modulations"),
}


def load_phy_tables(name):
    from scipy.io import loadmat
    fname, struct, _ = PHY_TABLES[name]
    d = loadmat(os.path.join(REF, "examples", "phy_frag_rmsa", "inputs", fname))[struct]
    return d[0][0][0], d[0][0][1], d[0][0][2]  # connections_detail, modulation_level, gsnr


def gen_phy_tables():
    """QoT tables as flat arrays: (source, destination) node numbers per row, modulation level uint8 and
    GSNR float64 per (row, channel, k-path) for the first k = 3 path columns (the shipped pickles have k = 3)."""
    os.makedirs(os.path.join(HERE, "tables"), exist_ok=True)
    for name in PHY_TABLES:
        conn, mod, gsnr = load_phy_tables(name)
        pairs = np.array([[int(np.asarray(r[0]).ravel()[0]), int(np.asarray(r[1]).ravel()[0])] for r in conn], np.int32)
        np.savez_compressed(os.path.join(HERE, "tables", name + ".npz"), pairs=pairs,
                            modulation_level=np.ascontiguousarray(mod[:, :, :3]),
                            gsnr=np.ascontiguousarray(gsnr[:, :, :3]))
        print("tables", name, pairs.shape, mod.shape, "levels", np.unique(mod[:, :, :3]).tolist())


def run_phy_trace(topo, tables, env_kwargs, policy, n_steps, reset_on_done):
    from optical_rl_gym.envs import phy_rmsa_env as P

    conn, mod, gsnr = tables
    env = P.PhyRMSAEnv(topology=topo, modulation_level=mod, connections_detail=conn, gsnr=gsnr, **env_kwargs)
    pol = {"bmfa": P.phy_aware_bmfa_rmsa, "bmfa_rss": P.phy_aware_bmfa_rss_rmsa, "sapff": P.sapff_rmsa,
           "bmff": P.phy_aware_bmff_rmsa, "sapbm": P.phy_aware_sapbm_rmsa, "faff": P.phy_aware_faff_rmsa,
           "faff_rss": P.phy_aware_faff_rss_rmsa}[policy]
    rec = Recorder()
    MAXCH = 12
    chans, used, free, cap = [], [], [], []
    for _ in range(n_steps):
        s = env.current_service
        a = pol(env)
        chosen = [tuple(c) for c in a[1]]  # the env keeps (and, when defragmenting, rewrites) the list object itself
        _, reward, done, _, info = env.step(a)
        row_c, row_u, row_f, row_k = [-1] * MAXCH, [0.0] * MAXCH, [0.0] * MAXCH, [0] * MAXCH
        assert len(chosen) <= MAXCH
        for i, c in enumerate(chosen):
            row_c[i], row_u[i], row_f[i], row_k[i] = int(c[0]), float(c[1]), float(c[2]), int(c[3])
        chans.append(row_c); used.append(row_u); free.append(row_f); cap.append(row_k)
        av = env.topology.graph["available_channels"]
        rec.add(
            service_id=s.service_id, src_id=s.source_id, dst_id=s.destination_id, bit_rate=s.bit_rate,
            arrival=s.arrival_time, holding=s.holding_time, act_path=int(a[0]), n_channels=len(a[1]),
            accepted=bool(s.accepted), virtual=bool(s.virtual_layer), reward=float(reward), done=bool(done),
            services_processed=env.services_processed, services_accepted=env.services_accepted,
            episode_services_processed=env.episode_services_processed,
            episode_services_accepted=env.episode_services_accepted,
            bit_rate_requested=env.bit_rate_requested, bit_rate_provisioned=env.bit_rate_provisioned,
            number_cuts_total=float(info["number_cuts_total"]), rss_total_metric=float(info["rss_total_metric"]),
            total_path_length=float(info["total_path_length"]), avrage_gsnr=float(info["avrage_gsnr"]),
            average_mod_level=float(info["average_mod_level"]), average_path_index=float(info["average_path_index"]),
            path_index=int(info["path_index"]), physical_paths=int(info["physical_paths"]),
            episode_service_blocking_rate=float(info["episode_service_blocking_rate"]),
            bit_rate_blocking_rate=float(info["bit_rate_blocking_rate"]),
            free_total=int(av.sum()), occ_crc=occ_crc(av), current_time=env.current_time,
            n_running=len(env.topology.graph["running_services"]),
            num_moves=float(info["num_moves"]), num_moves_groom=int(info["num_moves_groom"]),
            num_defrag_cycle=int(info["num_defrag_cycle"]),
        )
        if done and reset_on_done:
            env.reset()
    out = rec.arrays()
    out["channels"] = np.array(chans, np.int16)
    out["ch_used"] = np.array(used)
    out["ch_free"] = np.array(free)
    out["ch_cap"] = np.array(cap, np.int16)
    out["final_available_channels"] = np.packbits(
        env.topology.graph["available_channels"].astype(np.uint8), axis=1, bitorder="little")
    return out


PHY_BASE = dict(seed=10, allow_rejection=True, load=1400, mean_service_holding_time=25, episode_length=200,
                num_spectrum_resources=64, bit_rate_selection="discrete", number_spectrum_channels=80,
                number_spectrum_channels_s_band=108, grooming=False)

PHY_CASES = [
    # tests/test_rmsa_threads_us.py:133-148 (the live configuration: bmfa / bmfa_rss, grooming=False)
    ("phy_us14_s10_bmfa", "us14_k3", dict(), "bmfa", 800, True),
    ("phy_us14_s10_bmfa_rss", "us14_k3", dict(), "bmfa_rss", 600, True),
    ("phy_us14_s11_bmfa_load2400", "us14_k3", dict(seed=11, load=2400), "bmfa", 1500, True),
    ("phy_jpn12_s3_bmfa", "jpn12_k3", dict(seed=3, load=900), "bmfa", 800, True),
    ("phy_us14_s12_bmfa_load4000", "us14_k3", dict(seed=12, load=4000), "bmfa", 2600, True),  # reaches blocking
    # heuristics that always use the virtual (grooming) layer, for the next row of work
    ("phy_us14_s10_sapff", "us14_k3", dict(), "sapff", 600, True),
    ("phy_us14_s10_bmff", "us14_k3", dict(), "bmff", 600, True),
    ("phy_us14_s10_sapbm", "us14_k3", dict(), "sapbm", 600, True),
    # env.grooming=True: bmfa / bmfa_rss consult the virtual layer too (phy_rmsa_env.py:1377-1380, 1443-1446)
    ("phy_us14_s10_bmfa_groom", "us14_k3", dict(grooming=True), "bmfa", 800, True),
    ("phy_us14_s13_bmfa_rss_groom_load3000", "us14_k3", dict(seed=13, load=3000, grooming=True), "bmfa_rss", 1200, True),
    ("phy_us14_s14_sapff_load4000", "us14_k3", dict(seed=14, load=4000), "sapff", 2600, True),  # reaches blocking
    ("phy_jpn12_s5_bmff", "jpn12_k3", dict(seed=5, load=900), "bmff", 800, True),
    # periodic defragmentation (tests/test_rmsa_threads_us.py:190-251: defrag_period=10, number_moves=10, cut / rss)
    ("phy_us14_s10_bmfa_defrag_cut", "us14_k3", dict(defrag_period=10, number_moves=10), "bmfa", 600, True),
    ("phy_us14_s10_bmfa_rss_defrag_rss", "us14_k3", dict(defrag_period=10, number_moves=10, metric="rss"), "bmfa_rss", 450, True),
    ("phy_us14_s16_sapff_defrag_load3000", "us14_k3", dict(seed=16, load=3000, defrag_period=10, number_moves=10), "sapff", 700, True),
    ("phy_jpn12_s7_bmff_defrag_rss", "jpn12_k3", dict(seed=7, load=900, defrag_period=7, number_moves=4, metric="rss"), "bmff", 500, True),
    # fragmentation-aware first fit (tests/test_rmsa_threads_us.py:87-108)
    ("phy_us14_s10_faff", "us14_k3", dict(), "faff", 600, True),
    ("phy_us14_s15_faff_rss_load2400", "us14_k3", dict(seed=15, load=2400), "faff_rss", 800, True),
]


def gen_phy(only_missing=False, only_case=None):
    if only_case is None:
        gen_phy_tables()
    for name, tab, over, policy, steps, reset in PHY_CASES:
        if only_missing and os.path.exists(os.path.join(HERE, name + ".npz")):
            continue
        if only_case is not None and name != only_case:
            continue
        kw = dict(PHY_BASE)
        kw.update(over)
        topo = load_pickled_topology(TOPOLOGIES[PHY_TABLES[tab][2]])
        out = run_phy_trace(topo, load_phy_tables(tab), kw, policy, steps, reset)
        meta = dict(topology=PHY_TABLES[tab][2], tables=tab, env_kwargs=_jsonable(kw), policy=policy, steps=steps,
                    reset_on_done=reset)
        out["meta"] = np.array(json.dumps(meta))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "accepted", int(out["services_accepted"][-1]), "/", int(out["services_processed"][-1]),
              "virtual", int(out["virtual"].sum()), "max channels", int(out["n_channels"].max()),
              "running", int(out["n_running"].max()), "cuts", out["number_cuts_total"][-1])


# --------------------------------------------------------------------------- GN-model OSNR grid
def gen_phy_new():
    """Only the PhyRMSA cases whose fixture is not there yet."""
    gen_phy(only_missing=True)


def gen_osnr():
    """examples/calculate_osnr.py cannot be imported (it imports names that do not exist), has no caller
    and no test.  Its function body is self-contained arithmetic on duck-typed objects: the function text is executed
    here with the two broken import statements dropped, on SimpleNamespace inputs.  Only inputs and outputs are kept."""
    from types import SimpleNamespace as NS

    src = open(os.path.join(REF, "examples", "calculate_osnr.py")).read()
    keep = []
    for ln in src.splitlines():
        if ln.startswith("from optical_rl_gym.utils import"):
            continue  # imports Span / Link, which do not exist (utils.py:38-54 are commented out)
        if "prmsa_env" in ln and "import" in ln:
            ln = ln[: len(ln) - len(ln.lstrip())] + "pass"  # `from optical_rl_gym.envs.prmsa_env import PRMSAEnv`: no such module
        keep.append(ln)
    ns = {"Service": object}  # only used as an annotation in the signature
    exec(compile("\n".join(keep), "calculate_osnr_text", "exec"), ns)
    calc = ns["calculate_osnr"]

    rng = np.random.default_rng(2024)
    M = 300
    check_link_off, link_span_off, link_svc_off = [0], [0], [0]
    bandwidth, center, power = [], [], []
    span_len, span_att, span_nf = [], [], []
    svc_bw, svc_fc, svc_se, svc_self = [], [], [], []
    out = []
    grid_fc = 191.3e12 + 50e9 * np.arange(268)
    for m in range(M):
        nlinks = int(rng.integers(1, 8))
        cur = NS(service_id=1000, bandwidth=float(rng.choice([37.5e9, 50e9, 75e9])),
                 center_frequency=float(grid_fc[rng.integers(0, 268)]), launch_power=float(10 ** (rng.uniform(-3, 1) / 10) * 1e-3))
        links, topo = [], {}
        for li in range(nlinks):
            nspans = int(rng.integers(1, 12))
            att_db_km = float(rng.uniform(0.18, 0.25))
            spans = []
            for _ in range(nspans):
                length = float(rng.uniform(40, 80))
                spans.append(NS(length=length, attenuation_normalized=att_db_km / (2 * 10 * np.log10(np.exp(1)) * 1e3),
                                noise_figure_normalized=float(10 ** (rng.uniform(4.5, 6.5) / 10))))
            nsvc = int(rng.integers(0, 60)) if m % 7 else int(rng.integers(150, 267))
            chans = rng.choice(268, size=nsvc, replace=False)
            running = []
            for sid, c in enumerate(chans):
                f = float(grid_fc[c])
                if f == cur.center_frequency:
                    continue
                running.append(NS(service_id=sid, bandwidth=float(rng.choice([37.5e9, 50e9])), center_frequency=f,
                                  current_modulation=NS(spectral_efficiency=int(rng.integers(1, 7)))))
            if rng.random() < 0.8:  # the current service is usually in the link's running list too
                running.insert(int(rng.integers(0, len(running) + 1)), cur)
            n1, n2 = f"a{li}", f"b{li}"
            topo.setdefault(n1, {})[n2] = {"link": NS(spans=spans), "running_services": running}
            links.append(NS(node1=n1, node2=n2))
            for s in spans:
                span_len.append(s.length); span_att.append(s.attenuation_normalized); span_nf.append(s.noise_figure_normalized)
            link_span_off.append(len(span_len))
            for s in running:
                is_self = s is cur
                svc_bw.append(s.bandwidth); svc_fc.append(s.center_frequency)
                svc_se.append(1 if is_self else s.current_modulation.spectral_efficiency); svc_self.append(is_self)
            link_svc_off.append(len(svc_bw))
        check_link_off.append(len(link_span_off) - 1)
        cur.path = NS(links=links)
        bandwidth.append(cur.bandwidth); center.append(cur.center_frequency); power.append(cur.launch_power)
        out.append(float(calc(NS(topology=topo), cur)))
    np.savez_compressed(
        os.path.join(HERE, "osnr_grid.npz"),
        check_link_off=np.array(check_link_off, np.int32), link_span_off=np.array(link_span_off, np.int32),
        link_svc_off=np.array(link_svc_off, np.int32), bandwidth=np.array(bandwidth), center_frequency=np.array(center),
        launch_power=np.array(power), span_length_km=np.array(span_len), span_attenuation=np.array(span_att),
        span_noise_figure=np.array(span_nf), svc_bandwidth=np.array(svc_bw), svc_center_frequency=np.array(svc_fc),
        svc_se=np.array(svc_se, np.int32), svc_is_self=np.array(svc_self, np.uint8), gsnr_db=np.array(out))
    print("osnr grid:", M, "checks,", len(span_len), "spans,", len(svc_bw), "interferer entries; GSNR range",
          min(out), max(out))
    # SURVEY 8c known answer: 2 x 75 km spans, 1 interferer at +100 GHz, 50 GHz, 0 dBm
    att = 0.2 / (2 * 10 * np.log10(np.exp(1)) * 1e3)
    cur = NS(service_id=1, bandwidth=50e9, center_frequency=193.1e12, launch_power=1e-3)
    oth = NS(service_id=2, bandwidth=50e9, center_frequency=193.2e12, current_modulation=NS(spectral_efficiency=2))
    spans = [NS(length=75.0, attenuation_normalized=att, noise_figure_normalized=10 ** 0.55) for _ in range(2)]
    cur.path = NS(links=[NS(node1="a", node2="b")])
    print("known answer:", calc(NS(topology={"a": {"b": {"link": NS(spans=spans), "running_services": [cur, oth]}}}), cur))


# --------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--case", default=None, help="with --only phy: one PHY_CASES entry")
    args = ap.parse_args()
    install_gym_stub()
    import optical_rl_gym  # noqa: F401  (registers env ids)

    todo = [args.only] if args.only else ["topologies", "rmsa", "wrappers", "seed", "bookkeeping", "deeprmsa", "phy", "osnr"]
    for what in todo:
        fn = globals().get("gen_" + what)
        if what == "phy" and args.case:
            gen_phy(only_case=args.case)
            continue
        if fn is None:
            print("skip", what, "(generator not implemented yet)")
            continue
        fn()


if __name__ == "__main__":
    main()
