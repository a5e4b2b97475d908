"""Oracle DeepRMSA step / observation against golden traces of the reference (bit-exact)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, deeprmsa_to_rmsa_kwargs, load_golden, load_topology, oracle_env_from_kwargs

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "deeprmsa_*.npz")))


@pytest.mark.parametrize("case", CASES)
def test_deeprmsa_trace_bit_exact(case):
    z, meta = load_golden(case)
    topo = load_topology(meta["topology"])
    kw, j = deeprmsa_to_rmsa_kwargs(meta["env_kwargs"])
    env = oracle_env_from_kwargs(topo, kw, j=j, reward_mode=1)
    assert np.array_equal(env.observation(), z["obs0"])
    policy, actions = meta["policy"], None
    if policy == "random":
        policy, actions = "deeprmsa_external", z["action"]
    tr = env.run(policy, meta["steps"], reset_on_done=meta["reset_on_done"], actions=actions, with_obs=True)
    for f, g in (("src", "src_id"), ("dst", "dst_id"), ("bit_rate", "bit_rate"), ("accepted", "accepted"),
                 ("done", "done"), ("services_accepted", "services_accepted"), ("free_total", "free_total"),
                 ("act_slot", "act_slot"), ("episode_services_processed", "episode_services_processed")):
        assert np.array_equal(tr[f].astype(np.int64), z[g].astype(np.int64)), f
    for f, g in (("arrival", "arrival"), ("holding", "holding"), ("reward", "reward"),
                 ("network_compactness", "network_compactness"), ("current_time", "current_time")):
        assert np.array_equal(tr[f], z[g]), f
    bad = np.argwhere(tr["obs"] != z["obs"])
    assert bad.size == 0, (bad[:5], tr["obs"][tuple(bad[0])], z["obs"][tuple(bad[0])])
    av = env.available_slots()
    assert np.array_equal(np.packbits(av, axis=1, bitorder="little"), z["final_available_slots"])
