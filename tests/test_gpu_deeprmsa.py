"""GPU parity of the DeepRMSA step (action -> route/block -> RMSA step) and observation builder against
the reference's golden traces (env 0) and the oracle (all envs)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, deeprmsa_to_rmsa_kwargs, load_golden, load_topology, oracle_env_from_kwargs
from test_gpu_rmsa import device_log_in_oracle  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "deeprmsa_*.npz")))


STEP_KERNEL = "auto"


@pytest.fixture(autouse=True, params=["wave", "group"])
def step_kernel(request):
    """Both step kernels carry the DeepRMSA policies (include/orlg.h ORLG_KERNEL_*): every test runs against each."""
    global STEP_KERNEL
    STEP_KERNEL = request.param
    yield request.param
    STEP_KERNEL = "auto"


def make_env(topo, meta_kw, batch):
    from optical_rl_gym_amd import BatchedDeepRMSAEnv
    kw = dict(meta_kw)
    return BatchedDeepRMSAEnv(topo, batch, step_kernel=STEP_KERNEL, **kw)


@pytest.mark.parametrize("case", CASES)
def test_deeprmsa_golden(case, device_log_in_oracle):
    z, meta = load_golden(case)
    topo = load_topology(meta["topology"])
    n, batch = min(meta["steps"], 500), 4
    env = make_env(topo, meta["env_kwargs"], batch)
    kw, j = deeprmsa_to_rmsa_kwargs(meta["env_kwargs"])
    oracles = [oracle_env_from_kwargs(topo, kw, seed=kw["seed"] + i, j=j, reward_mode=1) for i in range(batch)]
    obs = env.observation()
    # observation equals the reference's bit for bit: it contains no time-derived value
    assert np.array_equal(obs[0], z["obs0"])
    for i, o in enumerate(oracles):
        assert np.array_equal(obs[i], o.observation()), i
    policy = meta["policy"]
    rng = np.random.default_rng(5)
    for t in range(n):
        if policy == "random":
            a = rng.integers(0, topo.k_paths * j + 1, batch).astype(np.int32)
            a[0] = z["action"][t]
            r = env.run("deeprmsa_external", 1, actions=a, auto_reset=True,
                        outputs=("act_path", "act_slot", "accepted", "reward", "done", "arrival"))
        else:
            r = env.run(policy, 1, auto_reset=True, outputs=("act_path", "act_slot", "accepted", "reward", "done", "arrival"))
        obs = env.observation()
        for i, o in enumerate(oracles):
            if policy == "random":
                ot = o.run("deeprmsa_external", 1, reset_on_done=True, actions=a[i:i + 1].copy())
            else:
                ot = o.run(policy, 1, reset_on_done=True)
            for f in ("act_path", "act_slot", "accepted", "reward", "done", "arrival"):
                assert r[f][0, i] == ot[f][0], (f, t, i, r[f][0, i], ot[f][0])
            oo = o.observation()
            assert np.array_equal(obs[i], oo), (t, i, np.nonzero(obs[i] != oo), obs[i], oo)
        # env 0 is the reference's own trace
        assert r["accepted"][0, 0] == z["accepted"][t] and r["act_slot"][0, 0] == z["act_slot"][t]
        assert r["reward"][0, 0] == z["reward"][t] and r["done"][0, 0] == z["done"][t]
        assert np.array_equal(obs[0], z["obs"][t]), (t, obs[0], z["obs"][t])
    for o in oracles:
        o.close()
    env.close()


def test_deeprmsa_observation_batch_32768(device_log_in_oracle):
    """SURVEY 8(d) config 4 / BASELINE configs[3] as written (B = 32 768, NSFNET S=320 j=1, holding 7.5, inter-arrival
    1/12, the DeepRMSA node request probabilities of the reference's tests/test_deeprmsa.py:30-47): observation build for the
    whole batch; spot checks against the oracle and structural properties for every env."""
    topo = load_topology("nsfnet_chen_5-paths_6-modulations")
    from conftest import DEEPRMSA_NODE_PROBS
    from optical_rl_gym_amd import BatchedDeepRMSAEnv
    kw = dict(j=1, mean_service_holding_time=7.5, mean_service_inter_arrival_time=1.0 / 12.0,
              node_request_probabilities=DEEPRMSA_NODE_PROBS, num_spectrum_resources=320, episode_length=50, seed=100)
    B = 32768
    env = BatchedDeepRMSAEnv(topo, B, step_kernel=STEP_KERNEL, **kw)
    env.run("deeprmsa_sap_ff", 300, auto_reset=True)
    obs = env.observation()
    assert obs.shape == (B, 54)
    req = env.requests()
    assert np.array_equal(obs[:, 0], req["bit_rate"] / 100)
    assert np.all(obs[:, 1:15].sum(axis=1) == 1) and np.all(obs[:, 15:29].sum(axis=1) == 1)
    assert np.array_equal(obs[:, 1:15].argmax(axis=1), np.minimum(req["src"], req["dst"]))
    assert np.array_equal(obs[:, 15:29].argmax(axis=1), np.maximum(req["src"], req["dst"]))
    okw, j = deeprmsa_to_rmsa_kwargs(kw)
    for i in (0, 5, 4097, 16384, 20000, 32767):
        o = oracle_env_from_kwargs(topo, okw, seed=100 + i, j=j, reward_mode=1)
        o.run("deeprmsa_sap_ff", 300, reset_on_done=True, fields=[])
        assert np.array_equal(obs[i], o.observation()), i
        o.close()
    # float32 observations (orlg_deeprmsa_observation_f32): every element the float64 value rounded once, i.e. what an agent's
    # obs.astype(np.float32) makes of the reference's Box(float64) vector
    obs32 = env.observation(dtype=np.float32)
    assert obs32.dtype == np.float32 and np.array_equal(obs32, obs.astype(np.float32))
    buf = np.full((B, 54), 7.0, np.float32)
    assert env.observation(out=buf) is buf and np.array_equal(buf, obs32)
    with pytest.raises(TypeError):
        env.observation(out=np.zeros((B, 54), np.float16))
    env.close()


@pytest.mark.parametrize("slots,j", [(400, 2), (200, 3), (64, 1), (512, 2)])
def test_deeprmsa_synthetic_shapes(tmp_path, slots, j, device_log_in_oracle):
    """Observation builder and block policies on shapes the golden traces do not have (seven words of slots on the eight-word
    layout, one word, j > 1 with few blocks): device vs oracle, observation and decisions bit for bit."""
    pytest.importorskip("networkx")
    import test_gpu_edge_cases as ec
    from optical_rl_gym_amd.topology_io import topology_from_txt
    rng = np.random.default_rng(slots + j)
    edges = ec._grid_edges(3, 3, rng)
    topo = topology_from_txt(ec._write_topology(tmp_path, "grid9", 9, edges), "grid9", k_paths=3)
    meta_kw = dict(j=j, mean_service_holding_time=10.0, mean_service_inter_arrival_time=10.0 / (0.25 * slots),
                   num_spectrum_resources=slots, episode_length=40, seed=slots)
    batch = 3
    env = make_env(topo, meta_kw, batch)
    kw, jj = deeprmsa_to_rmsa_kwargs(meta_kw)
    oracles = [oracle_env_from_kwargs(topo, kw, seed=kw["seed"] + i, j=jj, reward_mode=1) for i in range(batch)]
    for t in range(120):
        a = rng.integers(0, topo.k_paths * j + 1, batch).astype(np.int32)
        r = env.run("deeprmsa_external", 1, actions=a, auto_reset=True, outputs=("act_path", "act_slot", "accepted", "reward", "done"))
        obs = env.observation()
        for i, o in enumerate(oracles):
            ot = o.run("deeprmsa_external", 1, reset_on_done=True, actions=a[i:i + 1].copy())
            for f in ("act_path", "act_slot", "accepted", "reward", "done"):
                assert r[f][0, i] == ot[f][0], (f, t, i)
            assert np.array_equal(obs[i], o.observation()), (t, i)
    for o in oracles:
        o.close()
    env.close()


def test_deeprmsa_float32_observation_into_a_device_buffer():
    """orlg_deeprmsa_observation_f32 writing straight into a caller's device buffer (a torch tensor: the agent loop of
    bench.py) == the float64 observation rounded once.  In a child process: torch has to create its HIP context before the
    library does, and this test process has long initialised the library."""
    if STEP_KERNEL != "wave":
        pytest.skip("the observation kernel is the same for both step kernels")
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        import numpy as np, torch
        torch.zeros(1, device="cuda")
        sys.path[:0] = [%r, %r]
        from conftest import DEEPRMSA_NODE_PROBS, load_topology
        from optical_rl_gym_amd import BatchedDeepRMSAEnv
        B = 4099
        env = BatchedDeepRMSAEnv(load_topology("nsfnet_chen_5-paths_6-modulations"), B, num_spectrum_resources=320, j=1,
                                 mean_service_holding_time=7.5, mean_service_inter_arrival_time=1 / 12.0,
                                 node_request_probabilities=DEEPRMSA_NODE_PROBS, episode_length=50, seed=3)
        env.run("deeprmsa_sap_ff", 120, auto_reset=True)
        t64 = torch.zeros((B, env.obs_dim), dtype=torch.float64, device="cuda")
        t32 = torch.full((B, env.obs_dim), 7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        env.observation(out=t64); env.observation(out=t32)
        env.synchronize()
        a, b, h = t64.cpu().numpy(), t32.cpu().numpy(), env.observation()
        assert np.array_equal(a, h) and np.array_equal(b, h.astype(np.float32)) and b.dtype == np.float32
        # pinned host buffers are used in place (no staging copy, the call does not wait): observation rows written over the
        # bus, actions read over it -- against the pageable-buffer path of a twin environment
        twin = BatchedDeepRMSAEnv(load_topology("nsfnet_chen_5-paths_6-modulations"), B, num_spectrum_resources=320, j=1,
                                  mean_service_holding_time=7.5, mean_service_inter_arrival_time=1 / 12.0,
                                  node_request_probabilities=DEEPRMSA_NODE_PROBS, episode_length=50, seed=3)
        twin.run("deeprmsa_sap_ff", 120, auto_reset=True)
        p32 = torch.full((B, env.obs_dim), 7.0, dtype=torch.float32).pin_memory()
        acts = torch.from_numpy(np.random.default_rng(1).integers(0, 6, B).astype(np.int32)).pin_memory()
        for t in range(5):
            env.run("deeprmsa_external", 1, actions=acts, auto_reset=True)
            env.observation(out=p32)
            env.synchronize()
            twin.run("deeprmsa_external", 1, actions=acts.numpy().copy(), auto_reset=True)
            assert np.array_equal(p32.numpy(), twin.observation(dtype=np.float32)), t
        print("f32 device buffer ok")
    """) % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "f32 device buffer ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
