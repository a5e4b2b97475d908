"""PhyRMSAEnv single-env view driven by a heuristic callback on the reference's surface vs the reference's trace."""
import numpy as np
import pytest

from conftest import load_golden, load_phy_tables, load_topology

pytestmark = pytest.mark.gpu


def test_phy_view_bmfa_matches_reference():
    import optical_rl_gym_amd as pkg
    z, meta = load_golden("phy_us14_s10_bmfa")
    pairs, mod, gsnr = load_phy_tables(meta["tables"])
    env = pkg.PhyRMSAEnv(topology=load_topology(meta["topology"]), modulation_level=mod, connections_detail=pairs,
                         gsnr=gsnr, **meta["env_kwargs"])
    assert env.topology.graph["num_channel_resources"] == 268
    for t in range(90):
        s = env.current_service
        assert (s.source_id, s.destination_id, s.bit_rate) == (z["src_id"][t], z["dst_id"][t], z["bit_rate"][t])
        a = pkg.phy_aware_bmfa_rmsa(env)
        assert a[0] == z["act_path"][t] and len(a[1]) == z["n_channels"][t], t
        assert [c[0] for c in a[1]] == z["channels"][t][:len(a[1])].tolist()
        assert [c[1] for c in a[1]] == z["ch_used"][t][:len(a[1])].tolist()
        obs, reward, done, truncated, info = env.step(a)
        assert truncated is False and reward == z["reward"][t] and done == bool(z["done"][t])
        assert info["number_cuts_total"] == z["number_cuts_total"][t]
        assert info["rss_total_metric"] == z["rss_total_metric"][t]
        assert info["total_path_length"] == z["total_path_length"][t]
        np.testing.assert_allclose(info["avrage_gsnr"], z["avrage_gsnr"][t], rtol=1e-15)
        assert info["path_index"] == z["path_index"][t] and info["physical_paths"] == z["physical_paths"][t]
        assert info["episode_service_blocking_rate"] == z["episode_service_blocking_rate"][t]
        assert info["bit_rate_blocking_rate"] == z["bit_rate_blocking_rate"][t]
    env.close()


@pytest.mark.parametrize("case,heuristic,n", [("phy_us14_s10_sapff", "sapff_rmsa", 260), ("phy_us14_s10_bmff", "phy_aware_bmff_rmsa", 120),
                                              ("phy_us14_s10_sapbm", "phy_aware_sapbm_rmsa", 120),
                                              ("phy_us14_s10_bmfa_groom", "phy_aware_bmfa_rmsa", 120),
                                              ("phy_us14_s10_bmfa_rss", "phy_aware_bmfa_rss_rmsa", 60),
                                              ("phy_us14_s10_faff", "phy_aware_faff_rmsa", 80),
                                              ("phy_us14_s10_bmfa_defrag_cut", "phy_aware_bmfa_rmsa", 130)])
def test_phy_view_heuristics_with_virtual_layer(case, heuristic, n):
    """The heuristic callbacks on the single-env view (incl. use_existing_channels on env.channel_state and virtual-layer
    actions path = 20 + k-path) reproduce the reference's trace."""
    import optical_rl_gym_amd as pkg
    z, meta = load_golden(case)
    pairs, mod, gsnr = load_phy_tables(meta["tables"])
    env = pkg.PhyRMSAEnv(topology=load_topology(meta["topology"]), modulation_level=mod, connections_detail=pairs,
                         gsnr=gsnr, **meta["env_kwargs"])
    fn = getattr(pkg, heuristic)
    for t in range(n):
        a = fn(env)
        assert a[0] == z["act_path"][t] and len(a[1]) == z["n_channels"][t], t
        assert [c[0] for c in a[1]] == z["channels"][t][:len(a[1])].tolist()
        assert [c[1] for c in a[1]] == z["ch_used"][t][:len(a[1])].tolist()
        assert [c[2] for c in a[1]] == z["ch_free"][t][:len(a[1])].tolist()
        obs, reward, done, truncated, info = env.step(a)
        assert reward == z["reward"][t] and done == bool(z["done"][t])
        assert info["number_cuts_total"] == z["number_cuts_total"][t]
        assert info["physical_paths"] == z["physical_paths"][t]
        if "num_moves" in z.files:
            assert info["num_moves"] == z["num_moves"][t] and info["num_moves_groom"] == z["num_moves_groom"][t]
            assert info["num_defrag_cycle"] == z["num_defrag_cycle"][t]
        if done:
            env.reset()
    env.close()


def test_phy_view_bvt_counters():
    """PhyRMSAEnv.bvts (phy_rmsa_env.py:153-156, 603-608): one transceiver per channel a physical provisioning lights, by the band
    of its index and the service's node pair -- kept by the view on the host.  The reference's run (tests/golden/bookkeeping.npz:
    500 steps of phy_aware_bmff_rmsa, physical and virtual acceptances, episode resets) replayed through the view."""
    import json
    import os
    import optical_rl_gym_amd as pkg
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "bookkeeping.npz"))
    meta = json.loads(str(z["meta"]))["phy"]
    pairs, mod, gsnr = load_phy_tables(meta["tables"])
    env = pkg.PhyRMSAEnv(topology=load_topology(meta["topology"]), modulation_level=mod, connections_detail=pairs,
                         gsnr=gsnr, **meta["env_kwargs"])
    for t in range(meta["steps"]):
        a = pkg.phy_aware_bmff_rmsa(env)
        assert a[0] == z["phy_act_path"][t], t
        assert [c[0] for c in a[1]] == [c for c in z["phy_channels"][t].tolist() if c >= 0], t
        s = env.current_service
        _, _, done, _, _ = env.step(a)
        assert s.accepted == bool(z["phy_accepted"][t]), t
        if done:
            env.reset()
    assert env.bvts.shape == z["phy_bvts"].shape
    assert np.array_equal(env.bvts, z["phy_bvts"])
    env.close()
