"""Topology front-end (txt link list -> k shortest paths -> frozen tables) against the tables frozen from the
reference's shipped pickles (examples/topologies/*.h5)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_topology


@pytest.mark.parametrize("txt,k,frozen", [("nsfnet_chen", 5, "nsfnet_chen_5-paths_6-modulations"),
                                          ("us14", 3, "us14_3-paths_6-modulations"),
                                          ("jpn12", 3, "jpn12_3-paths_6-modulations"),
                                          ("jpn12", 5, "jpn12_5-paths_6-modulations")])
def test_front_end_reproduces_pickled_topologies(txt, k, frozen):
    pytest.importorskip("networkx")
    from optical_rl_gym_amd.topology_io import topology_from_txt
    want = load_topology(frozen)
    got = topology_from_txt(os.path.join(GOLDEN, "topology_txt", txt + ".txt"), want.name, k_paths=k)
    assert got.nodes == want.nodes and got.edges == want.edges
    assert np.array_equal(got.pair_path_base, want.pair_path_base)
    assert np.array_equal(got.path_hops, want.path_hops)
    assert np.array_equal(got.path_se, want.path_se)
    assert np.array_equal(got.path_length, want.path_length)
    assert np.array_equal(got.path_links, want.path_links)
