"""Randomised parity: random grid topologies / slot counts / k / loads / statistics-free device policies, device (both step
kernels) vs the oracle bit for bit on decisions, floats, occupancy, counters and link statistics (the helper of the edge-case
tests).  Seeds are fixed: the configurations are the same on every run."""
import numpy as np
import pytest

from test_gpu_rmsa import device_log_in_oracle  # noqa: F401
import test_gpu_edge_cases as ec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel", ["wave", "group"])
@pytest.mark.parametrize("case", range(24))
def test_random_configuration_vs_oracle(tmp_path, case, kernel, device_log_in_oracle):
    pytest.importorskip("networkx")
    from optical_rl_gym_amd.topology_io import topology_from_txt
    rng = np.random.default_rng(1000 + case)
    rows, cols = int(rng.integers(2, 5)), int(rng.integers(2, 5))
    edges = ec._grid_edges(rows, cols, rng)
    k, topo = int(rng.integers(1, 7)), None
    while topo is None:
        try:
            topo = topology_from_txt(ec._write_topology(tmp_path, f"g{case}", rows * cols, edges), f"g{case}", k_paths=k)
        except ValueError:   # a small grid does not have k simple paths for every pair
            k -= 1
    S = int(rng.choice([64, 100, 128, 200, 320, 400, 512]))
    kw = dict(num_spectrum_resources=S, load=max(2.0, float(rng.uniform(0.15, 0.6)) * S * len(edges) / 40.0),
              mean_service_holding_time=float(rng.uniform(5, 30)), episode_length=int(rng.integers(30, 200)),
              seed=int(rng.integers(1, 10000)))
    policy = str(rng.choice(["sap_ff", "sp_ff", "llp_ff"]))
    ec.STEP_KERNEL = kernel
    try:
        tr = ec._compare(topo, kw, policy, int(rng.integers(150, 400)), int(rng.choice([1, 3, 6])))
    finally:
        ec.STEP_KERNEL = "auto"
    assert tr["accepted"].shape[0] > 0
