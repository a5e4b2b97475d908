"""Checkpoint / resume of the batched state (orlg_save_state / orlg_load_state): a restored batch continues bit for bit."""
import numpy as np
import pytest

from conftest import load_golden, load_phy_tables, load_topology
from test_gpu_phy import make_env
from test_gpu_rmsa import make_batched

pytestmark = pytest.mark.gpu


def test_rmsa_save_load_resume(nsfnet):
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=100, seed=3)
    outs = ("act_path", "act_slot", "accepted", "arrival", "network_compactness")
    a = make_batched(nsfnet, kw, 40)
    a.run("sap_ff", 170, auto_reset=True)
    snap = a.save_state()
    want = a.run("sap_ff", 130, auto_reset=True, outputs=outs)
    a.load_state(snap)                                     # rewind the same handle
    again = a.run("sap_ff", 130, auto_reset=True, outputs=outs)
    b = make_batched(nsfnet, dict(kw, seed=999), 40)       # and restore into a fresh one
    b.load_state(snap)
    other = b.run("sap_ff", 130, auto_reset=True, outputs=outs)
    for f in outs:
        assert np.array_equal(want[f], again[f]) and np.array_equal(want[f], other[f]), f
    assert np.array_equal(a.available_slots(), b.available_slots())
    la, lb = a.link_stats(), b.link_stats()
    for name in la:
        assert np.array_equal(la[name], lb[name])
    a.close(); b.close()


def test_phy_save_load_resume():
    z, meta = load_golden("phy_us14_s10_bmfa_defrag_cut")
    topo, tables = load_topology(meta["topology"]), load_phy_tables(meta["tables"])
    kw = dict(meta["env_kwargs"], grooming=True)
    outs = ("act_path", "channels", "channels_used", "accepted", "number_cuts_total", "defrag_counters")
    a = make_env(topo, tables, kw, 6)
    a.run("bmfa", 140, auto_reset=True)
    snap = a.save_state()
    want = a.run("bmfa", 90, auto_reset=True, outputs=outs)
    b = make_env(topo, tables, dict(kw, seed=77), 6)
    b.load_state(snap)
    other = b.run("bmfa", 90, auto_reset=True, outputs=outs)
    for f in outs:
        assert np.array_equal(want[f], other[f]), f
    assert np.array_equal(a.available_channels(), b.available_channels())
    assert a.channel_state(3) == b.channel_state(3)
    a.close(); b.close()
