"""RMSA wrappers (SimpleMatrixObservation, PathOnlyFirstFitAction) on the device vs the reference's trace."""
import numpy as np
import pytest

from conftest import load_golden, load_topology, oracle_env_from_kwargs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("step_kernel", ["wave", "group"])
def test_wrappers_batched_and_view(step_kernel):
    import optical_rl_gym_amd as pkg
    z, meta = load_golden("wrappers_nsfnet_s21")
    topo = load_topology(meta["topology"])
    kw = {k: v for k, v in meta["env_kwargs"].items() if k != "allow_rejection"}
    B = 5
    env = pkg.BatchedRMSAEnv(topo, B, step_kernel=step_kernel, **kw)
    oracles = [oracle_env_from_kwargs(topo, meta["env_kwargs"], seed=kw["seed"] + i) for i in range(B)]
    rng = np.random.default_rng(9)
    for t in range(300):
        a = rng.integers(0, topo.k_paths + 1, B).astype(np.int32)
        a[0] = z["action"][t]
        r = env.run("path_ff_external", 1, actions=a, outputs=("act_path", "act_slot", "accepted", "reward"))
        obs = env.simple_matrix_observation()
        for i, o in enumerate(oracles):
            ot = o.run("path_ff_external", 1, actions=a[i:i + 1].copy())
            assert (r["act_path"][0, i], r["act_slot"][0, i], r["accepted"][0, i]) == \
                (ot["act_path"][0], ot["act_slot"][0], ot["accepted"][0]), (t, i)
            assert np.array_equal(obs[i], o.simple_matrix_observation().astype(np.uint8)), (t, i)
        assert (r["act_path"][0, 0], r["act_slot"][0, 0]) == (z["act_path"][t], z["act_slot"][t])
        assert r["reward"][0, 0] == z["reward"][t]
        assert np.array_equal(obs[0], z["obs"][t])
    env.close()
    # the same through the single-env wrapper classes
    w = pkg.PathOnlyFirstFitAction(pkg.SimpleMatrixObservation(pkg.RMSAEnv(topology=topo, **meta["env_kwargs"])))
    for t in range(60):
        assert w.action(int(z["action"][t])) == (z["act_path"][t], z["act_slot"][t])
        obs, reward, done, info = w.step(int(z["action"][t]))
        assert np.array_equal(obs, z["obs"][t]) and reward == z["reward"][t]
    w.close()
