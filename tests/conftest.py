import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_topology(name):
    from optical_rl_gym_amd import FrozenTopology

    return FrozenTopology.from_json(os.path.join(GOLDEN, "topologies", name + ".json"))


def topology_tables(topo):
    return dict(num_nodes=topo.num_nodes, num_links=topo.num_links, k_paths=topo.k_paths,
                pair_path_base=topo.pair_path_base, pair_path_count=topo.pair_path_count,
                path_hops=topo.path_hops, path_se=topo.path_se, path_length=topo.path_length,
                path_link_off=topo.path_link_off, path_links=topo.path_links)


DEFAULT_BIT_RATES = [200, 250, 300, 350, 400, 450, 500, 550, 600, 650, 700, 750, 800, 850, 900, 950, 1000,
                     1050, 1100, 1150, 1200]


# SURVEY 8(d) config 4 / BASELINE configs[3]: the DeepRMSA node request probabilities of the reference's own test
# (tests/test_deeprmsa.py:30-47) -- parameter values, NSFNET node order
DEEPRMSA_NODE_PROBS = [0.01801802, 0.04004004, 0.05305305, 0.01901902, 0.04504505, 0.02402402, 0.06706707, 0.08908909,
                       0.13813814, 0.12212212, 0.07607608, 0.12012012, 0.01901902, 0.16916917]


def oracle_env_from_kwargs(topo, env_kwargs, seed=None, j=1, reward_mode=0, asan=False):
    """Build an oracle env from reference-style RMSAEnv kwargs (rmsa_env.py:29-53)."""
    import oracle as orc
    from optical_rl_gym_amd import selection_tables

    kw = dict(env_kwargs)
    bit_rates = kw.get("bit_rates", DEFAULT_BIT_RATES)
    cont = kw.get("bit_rate_selection", "discrete") == "continuous"
    if cont:   # rmsa_env.py:95-101: rng.randint(lower, higher)
        bit_rates = list(range(int(kw.get("bit_rate_lower_bound", 25)), int(kw.get("bit_rate_higher_bound", 100)) + 1))
    _, src_cum, dst_cum, br_cum = selection_tables(kw.get("node_request_probabilities"),
                                                   None if cont else kw.get("bit_rate_probabilities"), topo.num_nodes, bit_rates)
    if cont:
        br_cum = None
    load, ht = kw.get("load", 10), kw.get("mean_service_holding_time", 10800.0)
    # optical_network_env.py:127-129 ; rmsa_env.py:646-651
    mean_iat = 1 / float(load / float(ht))
    return orc.OracleEnv(topology_tables(topo), num_slots=kw.get("num_spectrum_resources", 100),
                         episode_length=kw.get("episode_length", 1000), bit_rates=bit_rates, bit_rate_cum=br_cum,
                         src_cum=src_cum, dst_cum=dst_cum, arrival_lambda=1 / mean_iat, holding_lambda=1 / ht,
                         channel_width=kw.get("channel_width", 12.5), j=j, reward_mode=reward_mode,
                         seed=kw.get("seed", 41) if seed is None else seed, asan=asan)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return z, meta


@pytest.fixture(scope="session")
def nsfnet():
    return load_topology("nsfnet_chen_5-paths_6-modulations")


def deeprmsa_to_rmsa_kwargs(kw):
    """DeepRMSAEnv.__init__ (deeprmsa_env.py:10-32): load = holding / inter-arrival, S default 100."""
    kw = dict(kw)
    ht = kw.get("mean_service_holding_time", 25.0)
    iat = kw.pop("mean_service_inter_arrival_time", 0.1)
    j = kw.pop("j", 1)
    out = dict(load=ht / iat, mean_service_holding_time=ht, num_spectrum_resources=kw.get("num_spectrum_resources", 100),
               episode_length=kw.get("episode_length", 1000), seed=kw.get("seed"),
               node_request_probabilities=kw.get("node_request_probabilities"))
    return out, j


PHY_DEFAULT_BIT_RATES = [100, 200, 300, 400, 500, 600]  # phy_rmsa_env.py:38


def load_phy_tables(name):
    z = np.load(os.path.join(GOLDEN, "tables", name + ".npz"), allow_pickle=False)
    return z["pairs"], z["modulation_level"], z["gsnr"]


def phy_oracle_from_kwargs(topo, tables, env_kwargs, seed=None, asan=False):
    """Oracle PhyRMSAEnv from reference-style kwargs (phy_rmsa_env.py:30-58)."""
    import oracle as orc
    from optical_rl_gym_amd import selection_tables
    kw = dict(env_kwargs)
    pairs, mod, gsnr = tables
    bit_rates = kw.get("bit_rates", PHY_DEFAULT_BIT_RATES)
    _, src_cum, dst_cum, br_cum = selection_tables(kw.get("node_request_probabilities"),
                                                   kw.get("bit_rate_probabilities"), topo.num_nodes, bit_rates)
    load, ht = kw.get("load", 10), kw.get("mean_service_holding_time", 10800.0)
    mean_iat = 1 / float(load / float(ht))
    nch = 2 * kw.get("number_spectrum_channels", 80) + kw.get("number_spectrum_channels_s_band", 108)
    return orc.PhyOracleEnv(topology_tables(topo), num_channels=nch, episode_length=kw.get("episode_length", 1000),
                            bit_rates=bit_rates, bit_rate_cum=br_cum, src_cum=src_cum, dst_cum=dst_cum,
                            arrival_lambda=1 / mean_iat, holding_lambda=1 / ht,
                            pair_table_row=topo.pair_table_rows(pairs), modulation_level=mod, gsnr=gsnr,
                            link_ends=topo.link_ends, path_node_off=topo.path_node_off, path_nodes=topo.path_nodes,
                            grooming=kw.get("grooming", True), defrag_period=kw.get("defrag_period"),
                            number_moves=kw.get("number_moves"), metric=kw.get("metric", "cut"), seed=kw.get("seed", 41) if seed is None else seed, asan=asan,
                            gn_gate=kw.get("gn_gate"))
