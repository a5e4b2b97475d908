"""GPU parity of the QoT-aware (PhyRMSA) path -- physical and virtual layer, periodic defragmentation: the seven device
heuristics and external actions against the oracle (all envs, bit-exact with the device log in the oracle) and the reference's golden traces
(env 0: decisions / counters exactly, time-derived floats to rtol 1e-12)."""
import numpy as np
import pytest

from conftest import load_golden, load_phy_tables, load_topology, phy_oracle_from_kwargs
from test_gpu_rmsa import device_log_in_oracle  # noqa: F401

pytestmark = pytest.mark.gpu
OUTS = ("act_path", "n_channels", "channels", "channels_used", "accepted", "done", "request", "arrival", "holding",
        "number_cuts_total", "rss_total_metric", "defrag_counters")


def make_env(topo, tables, kw, batch, **extra):
    from optical_rl_gym_amd import BatchedPhyRMSAEnv
    pairs, mod, gsnr = tables
    kw = {k: v for k, v in kw.items() if k not in ("num_spectrum_resources", "bit_rate_selection")}
    return BatchedPhyRMSAEnv(topo, batch, modulation_level=mod, connections_detail=pairs, gsnr=gsnr, **kw, **extra)


@pytest.mark.parametrize("case,nmax", [("phy_us14_s10_bmfa", 800), ("phy_jpn12_s3_bmfa", 500),
                                       ("phy_us14_s12_bmfa_load4000", 2600), ("phy_us14_s10_bmfa_rss", 600),
                                       ("phy_us14_s10_sapff", 800), ("phy_us14_s10_bmff", 800),
                                       ("phy_us14_s10_sapbm", 800), ("phy_us14_s11_bmfa_load2400", 1500),
                                       ("phy_us14_s10_bmfa_groom", 800), ("phy_us14_s13_bmfa_rss_groom_load3000", 1200),
                                       ("phy_us14_s14_sapff_load4000", 2600), ("phy_jpn12_s5_bmff", 800),
                                       ("phy_us14_s10_faff", 600), ("phy_us14_s15_faff_rss_load2400", 800),
                                       ("phy_us14_s10_bmfa_defrag_cut", 600), ("phy_us14_s10_bmfa_rss_defrag_rss", 450),
                                       ("phy_us14_s16_sapff_defrag_load3000", 700), ("phy_jpn12_s7_bmff_defrag_rss", 500)])
def test_phy_policy_vs_oracle_and_reference(case, nmax, device_log_in_oracle, expect_node_vectors=None):
    z, meta = load_golden(case)
    if expect_node_vectors is None:   # the default for networks of at most 16 nodes (US14, JPN12): node-degree vectors
        expect_node_vectors = True
    topo = load_topology(meta["topology"])
    tables = load_phy_tables(meta["tables"])
    kw = meta["env_kwargs"]
    n, batch = min(nmax, meta["steps"]), 4
    env = make_env(topo, tables, kw, batch)
    assert env.grooming == kw.get("grooming", False)
    assert env.node_vectors == expect_node_vectors
    policy = meta["policy"]
    tr = env.run(policy, n, outputs=OUTS, auto_reset=True)
    cnt, now, nrun, av, est = env.counters(), env.current_time(), env.num_running(), env.available_channels(), env.episode_stats()
    for i in range(batch):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=kw["seed"] + i)
        ot = o.run(policy, n, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["n_channels"][:, i], ot["n_channels"]), i
        assert np.array_equal(tr["channels"][:, i, :12].astype(np.int32), ot["channels"]), i
        assert np.array_equal(tr["channels_used"][:, i, :12].astype(np.float64), ot["ch_used"]), i
        assert np.array_equal(tr["accepted"][:, i], ot["accepted"]) and np.array_equal(tr["done"][:, i], ot["done"])
        assert np.array_equal(tr["request"][:, i, 1], ot["src"]) and np.array_equal(tr["request"][:, i, 3], ot["bit_rate"])
        for f in ("arrival", "holding", "number_cuts_total", "rss_total_metric"):
            bad = np.nonzero(tr[f][:, i] != ot[f])[0]
            assert bad.size == 0, (f, i, bad[:4], tr[f][bad[:4], i], ot[f][bad[:4]])
        dc = tr["defrag_counters"][:, i].astype(np.int64)
        assert np.array_equal(dc[:, 1], ot["num_moves_groom"]) and np.array_equal(dc[:, 2], ot["num_defrag_cycle"]), i
        assert np.array_equal(dc[:, 0] / 2 + dc[:, 1], ot["num_moves"]), i
        oc = o.counters()
        for name in oc:
            assert cnt[name][i] == oc[name], (name, i)
        assert now[i] == o.current_time() and nrun[i] == o.num_running()
        assert np.array_equal(av[i], o.available_channels())
        # per-episode sums behind the info dict (the run ended right after an optional reset)
        assert est["physical_services_accepted"][i] == ot["physical_paths"][-1] or ot["done"][-1]
        assert est["queue_overflow"][i] == 0
        assert env.channel_state(i) == o.channel_state(), i
        o.close()
    # env 0 is the reference's own trace
    assert np.array_equal(tr["act_path"][:, 0], z["act_path"][:n])
    assert np.array_equal(tr["channels"][:, 0, :12], z["channels"][:n])
    assert np.array_equal(tr["channels_used"][:, 0, :12].astype(np.float64), z["ch_used"][:n])
    assert np.array_equal(tr["accepted"][:, 0], z["accepted"][:n])
    np.testing.assert_allclose(tr["arrival"][:, 0], z["arrival"][:n], rtol=1e-12)
    assert np.array_equal(tr["number_cuts_total"][:, 0], z["number_cuts_total"][:n])
    assert np.array_equal(tr["rss_total_metric"][:, 0], z["rss_total_metric"][:n])
    assert cnt["services_accepted"][0] == z["services_accepted"][n - 1]
    if "num_moves" in z.files:
        dc = tr["defrag_counters"][:, 0].astype(np.int64)
        assert np.array_equal(dc[:, 0] / 2 + dc[:, 1], z["num_moves"][:n])
        assert np.array_equal(dc[:, 1], z["num_moves_groom"][:n]) and np.array_equal(dc[:, 2], z["num_defrag_cycle"][:n])
    env.close()


@pytest.mark.parametrize("case,nmax", [("phy_us14_s10_bmfa", 500), ("phy_us14_s10_bmfa_defrag_cut", 400), ("phy_us14_s10_faff", 400),
                                       ("phy_jpn12_s5_bmff", 300)])
def test_phy_cut_metric_by_adjacency_lists(case, nmax, device_log_in_oracle, monkeypatch):
    """The cut metric has two evaluations on the device: byte dot products over per-node free degrees (networks of at most 16
    nodes: the default for US14 / JPN12) and the adjacency lists of calculate_r_cut itself (any network).  The suite above
    runs the first; ORLG_PHY_NODEVEC=0 forces the second, held to the same oracle and reference traces."""
    monkeypatch.setenv("ORLG_PHY_NODEVEC", "0")
    test_phy_policy_vs_oracle_and_reference(case, nmax, None, expect_node_vectors=False)


def test_phy_info_ratios_match_reference(device_log_in_oracle):
    """Step by step (one launch per step) the info-dict ratios equal the reference's."""
    z, meta = load_golden("phy_us14_s10_bmfa")
    topo, tables = load_topology(meta["topology"]), load_phy_tables(meta["tables"])
    env = make_env(topo, tables, meta["env_kwargs"], 2)
    for t in range(260):
        r = env.run("bmfa", 1, outputs=("done",))
        info = env.info()
        np.testing.assert_allclose(info["total_path_length"][0], z["total_path_length"][t], rtol=0, atol=0)
        np.testing.assert_allclose(info["avrage_gsnr"][0], z["avrage_gsnr"][t], rtol=1e-15)
        assert info["average_path_index"][0] == z["average_path_index"][t]
        assert info["path_index"][0] == z["path_index"][t] and info["physical_paths"][0] == z["physical_paths"][t]
        # NumPy-2 uint8 wrap of the reference's accumulator (SURVEY 8c caveat 2): reproduce it on the host
        st = env.episode_stats()[0]
        assert (st["total_modulation_level"] % 256) / (st["channels_accepted"] + 1) == z["average_mod_level"][t]
        if r["done"][0, 0]:
            env.reset()
    env.close()


def test_phy_external_actions(device_log_in_oracle):
    """External (path, channels) actions: replay the oracle's bmfa decisions, plus blocked and invalid actions."""
    z, meta = load_golden("phy_us14_s10_bmfa")
    topo, tables = load_topology(meta["topology"]), load_phy_tables(meta["tables"])
    kw = meta["env_kwargs"]
    env = make_env(topo, tables, kw, 3)
    oracles = [phy_oracle_from_kwargs(topo, tables, kw, seed=kw["seed"] + i) for i in range(3)]
    rng = np.random.default_rng(0)
    for t in range(150):
        paths = np.full(3, -2, np.int32)
        chans = np.full((3, 14), -1, np.int16)
        acts = []
        for i, o in enumerate(oracles):
            a = o.policy("bmfa")
            mode = rng.integers(0, 10)
            if mode == 0:      # proactively block
                a.path, a.n = -2, 0
            elif mode == 1 and t > 20:   # ask for channel 0..2 regardless of occupancy (often taken -> not accepted)
                a.n = 1
                a.ch[0] = int(rng.integers(0, 3))
            paths[i] = a.path
            for q in range(a.n):
                chans[i, q] = a.ch[q] | (int(a.used[q]) << 9)
            acts.append(a)
        r = env.run("external", 1, act_path=paths, act_channels=chans, outputs=("accepted", "arrival"))
        for i, o in enumerate(oracles):
            req = o.request()
            res = o.step(acts[i])
            assert r["accepted"][0, i] == res.accepted, (t, i)
            assert r["arrival"][0, i] == req.arrival_time
    av = env.available_channels()
    for i, o in enumerate(oracles):
        assert np.array_equal(av[i], o.available_channels())
        o.close()
    env.close()


@pytest.mark.parametrize("metric", ["cut", "rss"])
def test_phy_external_actions_with_defragmentation(metric, device_log_in_oracle):
    """The single-environment gym surface steps with external actions: the same with the periodic defragmentation inside the
    step (the external-action instantiation of the kernel + defragmentation), grooming on, per-step metrics written -- the
    oracle's own bmfa decisions replayed one launch per step, compared step by step."""
    topo, tables = load_topology("us14_3-paths_6-modulations"), load_phy_tables("us14_k3")
    kw = dict(load=1300, mean_service_holding_time=25, episode_length=90, seed=41, grooming=True, defrag_period=6, number_moves=5,
              metric=metric)
    env = make_env(topo, tables, kw, 3)
    oracles = [phy_oracle_from_kwargs(topo, tables, kw, seed=41 + i) for i in range(3)]
    policy = "bmfa" if metric == "cut" else "bmfa_rss"
    for t in range(260):
        paths = np.full(3, -2, np.int32)
        chans = np.full((3, 14), -1, np.int16)
        acts = []
        for i, o in enumerate(oracles):
            a = o.policy(policy)
            paths[i] = a.path
            for q in range(a.n):
                chans[i, q] = a.ch[q] | (int(a.used[q]) << 9)
            acts.append(a)
        r = env.run("external", 1, act_path=paths, act_channels=chans,
                    outputs=("accepted", "number_cuts_total", "rss_total_metric", "defrag_counters"), auto_reset=True)
        assert env.last_kernel().startswith("orlg_phy_kernel<5,true,false,-1>"), env.last_kernel()
        for i, o in enumerate(oracles):
            res = o.step(acts[i])
            assert r["accepted"][0, i] == res.accepted, (t, i)
            assert r["number_cuts_total"][0, i] == res.number_cuts_total, (t, i)
            assert r["rss_total_metric"][0, i] == res.rss_total_metric, (t, i)
            if res.done:
                o.reset(only_episode_counters=True)
    av, cnt = env.available_channels(), env.counters()
    for i, o in enumerate(oracles):
        assert np.array_equal(av[i], o.available_channels()), i
        assert cnt["services_accepted"][i] == o.counters()["services_accepted"], i
        assert env.channel_state(i) == o.channel_state(), i
        o.close()
    env.close()


def test_phy_external_virtual_layer_actions(device_log_in_oracle):
    """External actions incl. the virtual layer (path = 20 + k-path, per-channel shares): replay the oracle's sapff."""
    z, meta = load_golden("phy_us14_s10_sapff")
    topo, tables = load_topology(meta["topology"]), load_phy_tables(meta["tables"])
    kw = meta["env_kwargs"]
    env = make_env(topo, tables, kw, 2)
    oracles = [phy_oracle_from_kwargs(topo, tables, kw, seed=kw["seed"] + i) for i in range(2)]
    nvirt = 0
    for t in range(400):
        paths = np.full(2, -2, np.int32)
        chans = np.full((2, 14), -1, np.int16)
        acts = []
        for i, o in enumerate(oracles):
            a = o.policy("sapff")
            paths[i] = a.path
            nvirt += a.path > 10
            for q in range(a.n):
                chans[i, q] = a.ch[q] | (int(a.used[q]) << 9)
            acts.append(a)
        r = env.run("external", 1, act_path=paths, act_channels=chans, outputs=("accepted", "done"))
        for i, o in enumerate(oracles):
            res = o.step(acts[i])
            assert r["accepted"][0, i] == res.accepted, (t, i)
            if res.done:
                o.reset()
        if r["done"][0, 0]:
            env.reset()
    assert nvirt > 50
    av = env.available_channels()
    for i, o in enumerate(oracles):
        assert np.array_equal(av[i], o.available_channels())
        assert env.channel_state(i) == o.channel_state()
        o.close()
    env.close()


def test_phy_batch_4096_properties(device_log_in_oracle):
    """BASELINE config 3 size: B = 4096 PhyRMSA envs on US14 (load 1400, bmfa): size-independent properties for every env,
    sampled envs bit-exact against the oracle incl. the per-step cut / RSS metrics."""
    z, meta = load_golden("phy_us14_s10_bmfa")
    topo, tables = load_topology(meta["topology"]), load_phy_tables(meta["tables"])
    kw = dict(meta["env_kwargs"], seed=100)
    env = make_env(topo, tables, kw, 4096)
    tr = env.run("bmfa", 300, outputs=("accepted", "n_channels", "act_path", "channels", "number_cuts_total", "rss_total_metric"),
                 auto_reset=True)
    cnt = env.counters()
    av_all = env.available_channels()
    for i in (0, 1, 63, 64, 2047, 4095):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=100 + i)
        ot = o.run("bmfa", 300, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["channels"][:, i, :12].astype(np.int32), ot["channels"]), i
        assert np.array_equal(tr["number_cuts_total"][:, i], ot["number_cuts_total"]), i
        assert np.array_equal(tr["rss_total_metric"][:, i], ot["rss_total_metric"]), i
        assert np.array_equal(av_all[i], o.available_channels()), i
        assert cnt["services_accepted"][i] == o.counters()["services_accepted"], i
        o.close()
    assert np.all(cnt["services_processed"] == 301)
    assert np.array_equal(cnt["services_accepted"], tr["accepted"].sum(axis=0))
    av = env.available_channels()
    used = (1 - av.astype(np.int64)).sum(axis=(1, 2))
    assert np.all(used > 0) and np.all(env.num_running() <= 300)
    red, _ = env.reduce_counters()
    assert red["num_envs"] == 4096 and red["services_processed"] == 4096 * 301
    env.close()


def test_phy_work_queue_more_envs_than_resident_waves(device_log_in_oracle):
    """B = 9000 > resident waves: ticket mode for the long launches, static striding for the short ones; the near-term
    release buffer is rebuilt at every launch.  Sampled environments bit-exact against the oracle."""
    z, meta = load_golden("phy_us14_s10_sapff")
    topo, tables = load_topology(meta["topology"]), load_phy_tables(meta["tables"])
    kw = dict(meta["env_kwargs"], seed=500, load=600)
    B = 9000
    env = make_env(topo, tables, kw, B)
    env.run("sapff", 80, auto_reset=True)
    for _ in range(3):
        env.run("sapff", 4, auto_reset=True)
    tr = env.run("sapff", 60, outputs=("act_path", "channels", "accepted"), auto_reset=True)
    cnt, av = env.counters(), env.available_channels()
    assert env.episode_stats()["queue_overflow"].max() == 0
    for i in (0, 4095, 4096, 8191, 8192, 8999):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=500 + i)
        o.run("sapff", 92, reset_on_done=True, fields=[])
        ot = o.run("sapff", 60, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["channels"][:, i, :12].astype(np.int32), ot["channels"]), i
        assert np.array_equal(av[i], o.available_channels()), i
        assert cnt["services_accepted"][i] == o.counters()["services_accepted"], i
        assert env.channel_state(i) == o.channel_state(), i
        o.close()
    env.close()


@pytest.mark.parametrize("policy", ["bmfa", "bmfa_rss", "sapbm", "faff"])
def test_phy_five_paths_synthetic_tables(policy, device_log_in_oracle):
    """k = 5 candidate paths (the shipped QoT fixtures keep 3 columns): JPN12 with 5 paths and synthetic tables, grooming and
    defragmentation on, device vs oracle."""
    topo = load_topology("jpn12_5-paths_6-modulations")
    rng = np.random.default_rng(12)
    n = topo.num_nodes
    pairs = np.array([(a + 1, b + 1) for a in range(n) for b in range(a + 1, n)], np.int32)
    mod = rng.integers(1, 7, size=(len(pairs), 268, 5)).astype(np.uint8)
    gsnr = rng.uniform(5.0, 25.0, size=mod.shape)
    tables = (pairs, mod, gsnr)
    kw = dict(load=700, mean_service_holding_time=25, episode_length=120, seed=9, grooming=True, defrag_period=8, number_moves=5,
              metric="rss" if policy == "bmfa_rss" else "cut")
    env = make_env(topo, tables, kw, 3)
    tr = env.run(policy, 350, outputs=("act_path", "channels", "channels_used", "accepted", "number_cuts_total", "rss_total_metric",
                                       "defrag_counters"), auto_reset=True)
    av, cnt = env.available_channels(), env.counters()
    assert env.episode_stats()["queue_overflow"].max() == 0
    for i in range(3):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=9 + i)
        ot = o.run(policy, 350, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["channels"][:, i, :12].astype(np.int32), ot["channels"]), i
        assert np.array_equal(tr["channels_used"][:, i, :12].astype(np.float64), ot["ch_used"]), i
        assert np.array_equal(tr["number_cuts_total"][:, i], ot["number_cuts_total"]), i
        assert np.array_equal(tr["rss_total_metric"][:, i], ot["rss_total_metric"]), i
        assert np.array_equal(tr["defrag_counters"][:, i, 1], ot["num_moves_groom"]), i
        assert np.array_equal(av[i], o.available_channels()), i
        assert cnt["services_accepted"][i] == o.counters()["services_accepted"], i
        assert env.channel_state(i) == o.channel_state(), i
        o.close()
    assert (tr["act_path"] >= 3).any()   # paths 4 and 5 are used
    env.close()


@pytest.mark.parametrize("defrag", [None, "cut", "rss"])
@pytest.mark.parametrize("policy", ["bmfa", "bmfa_rss", "sapff", "bmff", "sapbm", "faff", "faff_rss"])
def test_phy_every_policy_instantiation_vs_oracle(policy, defrag, device_log_in_oracle):
    """The step kernel is instantiated per policy (x plain / + defragmentation): every one of them against the oracle on US14
    with the shipped tables, per-step metrics written, several launches (the node-degree vectors and the cached RSS terms are
    rebuilt at every launch start)."""
    topo, tables = load_topology("us14_3-paths_6-modulations"), load_phy_tables("us14_k3")
    kw = dict(load=1200, mean_service_holding_time=25, episode_length=150, seed=21, grooming=policy in ("bmfa", "bmfa_rss"))
    if defrag:
        kw.update(defrag_period=7, number_moves=6, metric=defrag)
    env = make_env(topo, tables, kw, 3)
    outs = ("act_path", "channels", "channels_used", "accepted", "number_cuts_total", "rss_total_metric", "defrag_counters")
    parts = [env.run(policy, n, outputs=outs, auto_reset=True) for n in (130, 1, 200, 69)]
    tr = {k: np.concatenate([q[k] for q in parts]) for k in outs}
    pol_index = {"bmfa": 0, "bmfa_rss": 1, "sapff": 2, "bmff": 3, "sapbm": 4, "faff": 5, "faff_rss": 6}[policy]
    assert env.last_kernel().startswith("orlg_phy_kernel<5,%s,%d>" % ("true,false" if defrag else "false,false", pol_index)), env.last_kernel()
    av, cnt = env.available_channels(), env.counters()
    assert env.episode_stats()["queue_overflow"].max() == 0
    for i in range(3):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=21 + i)
        ot = o.run(policy, 400, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["channels"][:, i, :12].astype(np.int32), ot["channels"]), i
        assert np.array_equal(tr["channels_used"][:, i, :12].astype(np.float64), ot["ch_used"]), i
        assert np.array_equal(tr["number_cuts_total"][:, i], ot["number_cuts_total"]), i
        assert np.array_equal(tr["rss_total_metric"][:, i], ot["rss_total_metric"]), i
        assert np.array_equal(tr["defrag_counters"][:, i, 1], ot["num_moves_groom"]), i
        assert np.array_equal(av[i], o.available_channels()), i
        assert cnt["services_accepted"][i] == o.counters()["services_accepted"], i
        assert env.channel_state(i) == o.channel_state(), i
        o.close()
    env.close()


@pytest.mark.parametrize("policy,launch_power_dbm,defrag", [("bmfa", 0.0, None), ("sapff", 2.0, None), ("faff", 1.0, None), ("bmff", 0.0, None),
                                                            ("bmfa_rss", 1.0, None), ("sapbm", 0.0, None), ("faff_rss", 2.0, None),
                                                            ("bmfa", 0.0, "cut"), ("sapff", 2.0, "rss"), ("faff", 1.0, "cut"),
                                                            ("bmff", 0.0, "rss"), ("bmfa_rss", 1.0, "rss"), ("sapbm", 0.0, "cut"),
                                                            ("faff_rss", 2.0, "rss")])
def test_gn_gate_in_the_step_vs_oracle(policy, launch_power_dbm, defrag, device_log_in_oracle):
    """GN-model admission check of the chosen channels inside the QoT-aware step (include/orlg.h orlg_gn_gate).  The reference
    gates by table only: this mode is PARITY UNPINNED by it and pinned to the oracle, which feeds its restatement of
    examples/calculate_osnr.py with the live occupancy.  Decisions, counters and occupancy must agree exactly, the GSNR values
    to rtol 1e-9 (the device sums the interferers of a link in a different order: ~1e-15); the gate must actually bind."""
    from optical_rl_gym_amd import gn_gate_parameters
    topo, tables = load_topology("us14_3-paths_6-modulations"), load_phy_tables("us14_k3")
    gate = gn_gate_parameters(topo, launch_power_dbm=launch_power_dbm)
    kw = dict(load=1400, mean_service_holding_time=25, episode_length=200, seed=10, grooming=False, gn_gate=gate)
    if defrag:   # the instantiation that carries both the defragmentation and the gate
        kw.update(defrag_period=10, number_moves=10, metric=defrag)
    n, batch = 700 if not defrag else 400, 4
    env = make_env(topo, tables, kw, batch)
    # a handle without defrag_period runs the gate's own instantiation (it does not carry the defragmentation's registers)
    assert env.last_kernel().startswith("orlg_phy_kernel<5,true,true," if defrag else "orlg_phy_kernel<5,false,true,"), env.last_kernel()
    tr = env.run(policy, n, outputs=("act_path", "channels", "accepted", "gn_gsnr_db", "number_cuts_total"), auto_reset=True)
    cnt, av = env.counters(), env.available_channels()
    gate_blocks = 0
    for i in range(batch):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=10 + i)
        ot = o.run(policy, n, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["channels"][:, i, :12].astype(np.int32), ot["channels"]), i
        assert np.array_equal(tr["accepted"][:, i], ot["accepted"]), i
        assert np.array_equal(tr["number_cuts_total"][:, i], ot["number_cuts_total"]), i
        g, og = tr["gn_gsnr_db"][:, i], ot["gn_gsnr_db"]
        assert np.array_equal(np.isnan(g), np.isnan(og)), i
        np.testing.assert_allclose(g[~np.isnan(g)], og[~np.isnan(og)], rtol=1e-9, atol=0)
        assert np.array_equal(av[i], o.available_channels()), i
        oc = o.counters()
        for name in oc:
            assert cnt[name][i] == oc[name], (name, i)
        gate_blocks += int(((ot["act_path"] >= 0) & (ot["act_path"] < 10) & (ot["accepted"] == 0)).sum())
        o.close()
    assert gate_blocks > (20 if not defrag else 8)   # the gate rejects some of the table-approved choices ...
    assert tr["accepted"].mean() > 0.3  # ... and passes others
    # a virtual-layer service lights nothing new: no check
    virt = tr["act_path"] >= 20
    assert np.all(np.isnan(tr["gn_gsnr_db"][virt]))
    env.close()


def test_gn_gate_batch_4096(device_log_in_oracle):
    """BASELINE configs[2] as worded -- "QoT-aware RMSA on USNET (GN-model OSNR gate + modulation-format selection), batch =
    4 096": US14, load 1400, bmfa with the GN-model admission check of the chosen channels inside the step
    (examples/calculate_osnr.py:9-56 against the live occupancy; PARITY UNPINNED by the reference, which gates by table only:
    phy_rmsa_env.py:596).  Eight sampled environments -- first / last of the batch, of a workgroup (8 waves), of the resident
    set -- against the oracle: decisions, occupancy and counters exactly, the GSNR of every check to rtol 1e-9; then the
    size-independent properties for every environment."""
    from optical_rl_gym_amd import gn_gate_parameters
    topo, tables = load_topology("us14_3-paths_6-modulations"), load_phy_tables("us14_k3")
    gate = gn_gate_parameters(topo)
    kw = dict(load=1400, mean_service_holding_time=25, episode_length=200, seed=10, grooming=False, gn_gate=gate)
    n, batch = 320, 4096
    env = make_env(topo, tables, kw, batch)
    tr = env.run("bmfa", n, outputs=("act_path", "channels", "accepted", "gn_gsnr_db", "number_cuts_total"), auto_reset=True)
    assert env.last_kernel().startswith("orlg_phy_kernel<5,false,true,"), env.last_kernel()
    cnt, av = env.counters(), env.available_channels()
    gate_blocks = 0
    for i in (0, 7, 8, 2047, 2048, 4088, 4094, 4095):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=10 + i)
        ot = o.run("bmfa", n, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["channels"][:, i, :12].astype(np.int32), ot["channels"]), i
        assert np.array_equal(tr["accepted"][:, i], ot["accepted"]), i
        assert np.array_equal(tr["number_cuts_total"][:, i], ot["number_cuts_total"]), i
        g, og = tr["gn_gsnr_db"][:, i], ot["gn_gsnr_db"]
        assert np.array_equal(np.isnan(g), np.isnan(og)), i
        np.testing.assert_allclose(g[~np.isnan(g)], og[~np.isnan(og)], rtol=1e-9, atol=0)
        assert np.array_equal(av[i], o.available_channels()), i
        oc = o.counters()
        for name in oc:
            assert cnt[name][i] == oc[name], (name, i)
        gate_blocks += int(((ot["act_path"] >= 0) & (ot["act_path"] < 10) & (ot["accepted"] == 0)).sum())
        o.close()
    assert gate_blocks > 20
    assert np.all(cnt["services_processed"] == n + 1)
    # (accepted of the running episode vs the launch's outputs: an episode of 200 ended inside the launch)
    checked = ~np.isnan(tr["gn_gsnr_db"])
    assert checked.mean() > 0.5 and np.all(tr["gn_gsnr_db"][checked] > 0) and np.all(tr["gn_gsnr_db"][checked] < 40)
    red, _ = env.reduce_counters()
    assert red["num_envs"] == batch and red["services_processed"] == batch * (n + 1)
    assert env.episode_stats()["queue_overflow"].max() == 0
    env.close()


def test_phy_reseed_between_launches_vs_oracle(device_log_in_oracle):
    """orlg_phy_reseed (a fresh generator for all five draws of a request: see orlg_reseed in include/orlg.h) between launches,
    with the defragmentation running, against the oracle's reseed()."""
    topo, tables = load_topology("us14_3-paths_6-modulations"), load_phy_tables("us14_k3")
    kw = dict(load=1400, mean_service_holding_time=25, episode_length=200, seed=10, grooming=True, defrag_period=10, number_moves=10)
    env = make_env(topo, tables, kw, 4)
    outs = ("act_path", "channels", "accepted", "arrival", "request")
    parts = [env.run("bmfa", 130, outputs=outs, auto_reset=True)]
    env.reseed(901)
    parts.append(env.run("bmfa", 270, outputs=outs, auto_reset=True))
    for i in range(4):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=10 + i)
        op = [o.run("bmfa", 130, reset_on_done=True)]
        o.reseed(901 + i)
        op.append(o.run("bmfa", 270, reset_on_done=True))
        for a, b in zip(parts, op):
            assert np.array_equal(a["act_path"][:, i], b["act_path"]), i
            assert np.array_equal(a["channels"][:, i, :12].astype(np.int32), b["channels"]), i
            assert np.array_equal(a["accepted"][:, i], b["accepted"]) and np.array_equal(a["arrival"][:, i], b["arrival"]), i
            assert np.array_equal(a["request"][:, i, 3], b["bit_rate"]), i
        assert np.array_equal(env.available_channels()[i], o.available_channels()), i
        o.close()
    env.close()


@pytest.mark.parametrize("policy,defrag", [("bmfa", None), ("bmfa_rss", None), ("sapff", "cut"), ("bmfa", "rss"), ("faff", "cut")])
def test_phy_large_network_general_paths(policy, defrag, device_log_in_oracle):
    """SPN (30 nodes, 56 links) is beyond the fast paths' structural limits: no node-degree vectors (N > 16), no
    32-bit link-axis columns and no incremental per-step totals (E > 32) -- the cut metric walks the adjacency lists, the RSS
    metric the link axis, number_cuts_total / rss_total_metric are rebuilt from all columns every step, the defragmentation
    scores through the same general code.  Synthetic QoT tables (the reference ships none for SPN), device vs oracle."""
    topo = load_topology("spn_3-paths_6-modulations")
    rng = np.random.default_rng(3)
    n = topo.num_nodes
    pairs = np.array([(a + 1, b + 1) for a in range(n) for b in range(a + 1, n)], np.int32)
    mod = rng.integers(1, 7, size=(len(pairs), 268, 3)).astype(np.uint8)
    gsnr = rng.uniform(5.0, 25.0, size=mod.shape)
    tables = (pairs, mod, gsnr)
    kw = dict(load=2500, mean_service_holding_time=25, episode_length=150, seed=4, grooming=policy != "sapff")
    if defrag:
        kw.update(defrag_period=9, number_moves=7, metric=defrag)
    env = make_env(topo, tables, kw, 3)
    assert not env.node_vectors
    outs = ("act_path", "channels", "channels_used", "accepted", "number_cuts_total", "rss_total_metric", "defrag_counters")
    parts = [env.run(policy, k, outputs=outs, auto_reset=True) for k in (180, 2, 120)]
    tr = {k: np.concatenate([q[k] for q in parts]) for k in outs}
    av, cnt = env.available_channels(), env.counters()
    assert env.episode_stats()["queue_overflow"].max() == 0
    for i in range(3):
        o = phy_oracle_from_kwargs(topo, tables, kw, seed=4 + i)
        ot = o.run(policy, 302, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["channels"][:, i, :12].astype(np.int32), ot["channels"]), i
        assert np.array_equal(tr["channels_used"][:, i, :12].astype(np.float64), ot["ch_used"]), i
        assert np.array_equal(tr["number_cuts_total"][:, i], ot["number_cuts_total"]), i
        assert np.array_equal(tr["rss_total_metric"][:, i], ot["rss_total_metric"]), i
        dc = tr["defrag_counters"][:, i].astype(np.int64)
        assert np.array_equal(dc[:, 1], ot["num_moves_groom"]) and np.array_equal(dc[:, 0] / 2 + dc[:, 1], ot["num_moves"]), i
        assert np.array_equal(av[i], o.available_channels()), i
        assert cnt["services_accepted"][i] == o.counters()["services_accepted"], i
        assert env.channel_state(i) == o.channel_state(), i
        o.close()
    assert tr["accepted"].mean() > 0.5
    env.close()
