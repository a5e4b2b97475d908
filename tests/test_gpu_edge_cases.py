"""Edge cases of the RMSA path on synthetic topologies (built with the package's own front-end from a link list), device vs
oracle bit for bit: the limits of the lane layout (k * W = 64 with S = 512, many links), a two-node network, slot counts that
are not a multiple of 64, and an over-provisioned queue."""
import numpy as np
import pytest

from conftest import oracle_env_from_kwargs
from test_gpu_rmsa import device_log_in_oracle, make_batched  # noqa: F401

pytestmark = pytest.mark.gpu


def _write_topology(tmp_path, name, num_nodes, edges):
    path = tmp_path / (name + ".txt")
    with open(path, "w") as f:
        f.write(f"{num_nodes}\n{len(edges)}\n")
        for a, b, length in edges:
            f.write(f"{a} {b} {length}\n")
    return str(path)


def _grid_edges(rows, cols, rng):
    edges = []
    node = lambda r, c: r * cols + c + 1
    for r in range(rows):
        for c in range(cols):
            if c + 1 < cols:
                edges.append((node(r, c), node(r, c + 1), int(rng.integers(60, 400))))
            if r + 1 < rows:
                edges.append((node(r, c), node(r + 1, c), int(rng.integers(60, 400))))
            if r + 1 < rows and c + 1 < cols and (r + c) % 2 == 0:
                edges.append((node(r, c), node(r + 1, c + 1), int(rng.integers(80, 500))))
    return edges


STEP_KERNEL = "auto"


@pytest.fixture(autouse=True, params=["wave", "group"])
def step_kernel(request):
    """Every edge case runs against both step kernels (the four-environments-per-wave kernel carries llp_ff itself since round 2;
    a shape whose four environments do not fit its LDS budget is served by the other one as well)."""
    global STEP_KERNEL
    STEP_KERNEL = request.param
    yield request.param
    STEP_KERNEL = "auto"


def _compare(topo, kw, policy, n, batch, outs=("act_path", "act_slot", "accepted", "arrival", "network_compactness")):
    from optical_rl_gym_amd import OrlgError
    try:
        env = make_batched(topo, kw, batch, step_kernel=STEP_KERNEL)
    except OrlgError as e:
        if STEP_KERNEL == "group" and "LDS" in str(e):
            pytest.skip("four environments of this shape do not fit the LDS: the wave-per-environment kernel serves it")
        raise
    tr = env.run(policy, n, outputs=outs, auto_reset=True)
    occ, cnt = env.available_slots(), env.counters()
    ls = env.link_stats()
    for i in range(batch):
        o = oracle_env_from_kwargs(topo, kw, seed=kw["seed"] + i)
        ot = o.run(policy, n, reset_on_done=True)
        for f in outs:
            assert np.array_equal(tr[f][:, i], ot[f]), (f, i)
        assert np.array_equal(occ[i], o.available_slots()), i
        oc = o.counters()
        for name in oc:
            assert cnt[name][i] == oc[name], (name, i)
        ols = o.link_stats()
        for name in ols:
            assert np.array_equal(ls[name][i], ols[name]), (name, i)
        o.close()
    env.close()
    return tr


@pytest.mark.parametrize("policy", ["sap_ff", "llp_ff"])
def test_widest_layout_k8_s512(tmp_path, policy, device_log_in_oracle):
    """k = 8 paths x W = 8 words fill the 64 lanes; 5 x 6 grid with diagonals: 30 nodes, 64 links."""
    pytest.importorskip("networkx")
    from optical_rl_gym_amd.topology_io import topology_from_txt
    rng = np.random.default_rng(3)
    edges = _grid_edges(5, 6, rng)
    topo = topology_from_txt(_write_topology(tmp_path, "grid30", 30, edges), "grid30", k_paths=8)
    assert topo.num_links == len(edges) and int(topo.path_hops.max()) <= 14
    kw = dict(num_spectrum_resources=512, load=400, mean_service_holding_time=20, episode_length=150, seed=31)
    tr = _compare(topo, kw, policy, 400, 3)
    assert 0 < tr["accepted"].mean() < 1  # the load reaches blocking


def test_two_node_network(tmp_path, device_log_in_oracle):
    pytest.importorskip("networkx")
    from optical_rl_gym_amd.topology_io import topology_from_txt
    topo = topology_from_txt(_write_topology(tmp_path, "pair", 2, [(1, 2, 300)]), "pair", k_paths=1)
    kw = dict(num_spectrum_resources=64, load=12, mean_service_holding_time=10, episode_length=50, seed=5)
    _compare(topo, kw, "sap_ff", 300, 4)


@pytest.mark.parametrize("slots", [65, 100, 191, 320, 384, 400, 448])   # 400 / 448: seven words of slots on the eight-word layout
def test_slot_counts_off_the_word_boundary(tmp_path, slots, device_log_in_oracle):
    pytest.importorskip("networkx")
    from optical_rl_gym_amd.topology_io import topology_from_txt
    rng = np.random.default_rng(slots)
    edges = _grid_edges(3, 3, rng)
    topo = topology_from_txt(_write_topology(tmp_path, "grid9", 9, edges), "grid9", k_paths=3)
    kw = dict(num_spectrum_resources=slots, load=30 * slots / 100, mean_service_holding_time=15, episode_length=120, seed=slots)
    _compare(topo, kw, "sap_ff", 300, 2)
