"""Full-size batches of BASELINE.json on the device, held to the oracle on sampled environments:
the north-star batch (65 536 environments, four-environments-per-wave kernel picked by AUTO) and the topology groups of
configs[4] (NSFNET / JPN12 / US14) at 32 768 environments each, the per-GPU batch of that configuration."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_topology, oracle_env_from_kwargs

pytestmark = pytest.mark.gpu


@pytest.fixture()
def device_log_in_oracle():
    import oracle as orc
    from optical_rl_gym_amd import _lib
    orc.set_log_fn(C.cast(_lib.load().orlg_host_log, C.c_void_p).value)
    yield
    orc.set_log_fn(None)


def _check_samples(env, topo, kw, policy, warm, n, tr, samples):
    occ_w = env.occupancy_words()
    now, cnt = env.current_time(), env.counters()
    ls = env.link_stats()
    S = kw["num_spectrum_resources"]
    for i in samples:
        o = oracle_env_from_kwargs(topo, kw, seed=kw["seed"] + i)
        o.run(policy, warm, fields=[])
        ot = o.run(policy, n)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]) and np.array_equal(tr["act_slot"][:, i], ot["act_slot"]), i
        assert np.array_equal(tr["accepted"][:, i], ot["accepted"]), i
        bits = np.unpackbits(occ_w[i].view(np.uint8), axis=-1, bitorder="little")[:, :S]
        assert np.array_equal(bits, o.available_slots()), i
        assert now[i] == o.current_time(), i
        oc = o.counters()
        for name in oc:
            assert cnt[name][i] == oc[name], (name, i)
        ols = o.link_stats()
        for name in ols:
            assert np.array_equal(ls[name][i], ols[name]), (name, i)
        o.close()


def test_north_star_batch_65536(device_log_in_oracle):
    """B = 65 536 on NSFNET-320 (the batch BASELINE.json's north_star quotes its target on, bench.py's headline): AUTO must
    pick the four-environments-per-wave kernel; a long launch (ticket queue over 16 384 quads), short launches (static
    striding) and a long launch with outputs; 12 sampled environments -- first / last of the batch, of a quad, of the
    resident set -- bit-exact against the oracle on decisions, occupancy, clock, counters and link statistics."""
    from optical_rl_gym_amd import BatchedRMSAEnv
    topo = load_topology("nsfnet_chen_5-paths_6-modulations")
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=10)
    B = 65536
    env = BatchedRMSAEnv(topo, B, **kw)
    env.run("sap_ff", 150)
    # (150 steps, full statistics, no per-step link outputs: the instantiation that defers the links' float64 updates)
    assert env.last_kernel().startswith("orlg_rmsa_group_kernel<5,2,false,true>"), env.last_kernel()
    for _ in range(2):
        env.run("sap_ff", 3)
    tr = env.run("sap_ff", 120, outputs=("act_path", "act_slot", "accepted"))
    cnt = env.counters()
    assert np.all(cnt["services_processed"] == 277)
    assert np.array_equal(cnt["services_accepted"] >= tr["accepted"].sum(axis=0), np.ones(B, bool))
    samples = (0, 1, 2, 3, 4, 11263, 11264, 16383, 32768, 50001, 65532, 65535)
    _check_samples(env, topo, kw, "sap_ff", 156, 120, tr, samples)
    red, _ = env.reduce_counters()
    assert red["num_envs"] == B and red["services_processed"] == 277 * B
    assert red["services_accepted"] == int(cnt["services_accepted"].sum())
    env.close()


@pytest.mark.parametrize("name", ["nsfnet_chen_5-paths_6-modulations", "jpn12_5-paths_6-modulations", "us14_3-paths_6-modulations"])
def test_mixed_topology_groups_32768(name, device_log_in_oracle):
    """The three topology groups of BASELINE configs[4] (JPN12 stands in for "JPN48": SURVEY 0.7) at that configuration's
    per-GPU batch, 262 144 / 8 = 32 768 environments, on the kernel AUTO picks for it (four environments per wave); sampled
    environments -- first / last of the batch, of a quad, of the resident set, of the ticket queue -- against the oracle."""
    from optical_rl_gym_amd import BatchedRMSAEnv
    topo = load_topology(name)
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=10)
    B = 32768
    env = BatchedRMSAEnv(topo, B, **kw)
    env.run("sap_ff", 200)
    tr = env.run("sap_ff", 150, outputs=("act_path", "act_slot", "accepted"))
    assert env.last_kernel().startswith("orlg_rmsa_group_kernel"), env.last_kernel()
    _check_samples(env, topo, kw, "sap_ff", 200, 150, tr, (0, 3, 4, 4095, 4096, 11263, 11264, 16384, 20001, 32764, 32767))
    red, _ = env.reduce_counters()
    assert red["num_envs"] == B and red["services_processed"] == 351 * B
    env.close()


@pytest.mark.parametrize("B,n,outs", [(20004, 400, ("act_path", "act_slot", "accepted", "done")), (65536, 1000, ("accepted",))])
def test_tickets_in_chunks_of_steps(device_log_in_oracle, B, n, outs):
    """Long launches of batches beyond one round of resident waves hand a quad's launch out in CHUNKS of steps (the work queue of
    orlg_rmsa_group_kernel: ticket = (quad, chunk); the wave that draws a chunk waits for the quad's previous chunk to be
    published -- agent-scope release / acquire, the two waves may sit on different XCDs).  B = 20 004 (5 001 quads on 3 072 wave
    slots: every wave hands quads over, under the uneven load of a last partial round; a partial last quad), three launches of
    400 steps with per-step outputs (and the headline's batch, B = 65 536, three launches of 1000 steps), against the same launches with whole-launch tickets (ORLG_NO_CHUNKS): outputs at their
    step's row, counters, link statistics and the saved state byte for byte; forced to 7 chunks as well; spot checks against
    the oracle."""
    import os
    from optical_rl_gym_amd import BatchedRMSAEnv
    nsfnet = load_topology("nsfnet_chen_5-paths_6-modulations")
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=300, seed=901)

    def drive(env_vars):
        old = {k: os.environ.get(k) for k in ("ORLG_NO_CHUNKS", "ORLG_GROUP_CHUNKS")}
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env_vars)
        try:
            env = BatchedRMSAEnv(nsfnet, B, step_kernel="group", **kw)
            runs = [env.run("sap_ff", n, outputs=outs, auto_reset=True) for _ in range(3)]
            name = env.last_kernel()
            res = (runs, env.save_state().copy(), {k: v.copy() for k, v in env.counters().items()},
                   {k: v.copy() for k, v in env.link_stats().items()}, name)
            env.close()
            return res
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v

    ref = drive({"ORLG_NO_CHUNKS": "1"})
    assert ref[4].endswith("chunks=1"), ref[4]
    for env_vars in ({}, {"ORLG_GROUP_CHUNKS": "7"}):
        got = drive(env_vars)
        assert not got[4].endswith("chunks=1"), got[4]
        for x, y in zip(got[0], ref[0]):
            for k in outs:
                assert np.array_equal(x[k], y[k]), (env_vars, k)
        assert np.array_equal(got[1], ref[1]), env_vars
        for k in ref[2]:
            assert np.array_equal(got[2][k], ref[2][k]), (env_vars, k)
        for k in ref[3]:
            assert np.array_equal(got[3][k], ref[3][k]), (env_vars, k)
    for i in (0, 4999, 12345, B - 1):
        o = oracle_env_from_kwargs(nsfnet, kw, seed=901 + i)
        tr = o.run("sap_ff", 3 * n, reset_on_done=True)
        got_acc = np.concatenate([r["accepted"][:, i] for r in ref[0]])
        assert np.array_equal(got_acc, tr["accepted"]), i
        o.close()
