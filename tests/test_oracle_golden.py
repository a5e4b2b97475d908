"""The CPU oracle (oracle/orlg_oracle.c) against golden vectors recorded from the unmodified
reference (tests/golden/make_golden.py).  Everything is compared bit-for-bit, floats included."""
import glob
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, load_topology, oracle_env_from_kwargs

RMSA_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "rmsa_*.npz")))

INT_FIELDS = ["service_id", "bit_rate", "accepted", "done", "services_processed", "services_accepted",
              "episode_services_processed", "episode_services_accepted", "bit_rate_requested",
              "bit_rate_provisioned", "episode_bit_rate_requested", "episode_bit_rate_provisioned", "free_total"]
FLOAT_FIELDS = ["arrival", "holding", "reward", "network_compactness", "network_compactness_difference",
                "avg_link_compactness", "avg_link_utilization", "fairness", "current_time", "graph_throughput",
                "graph_compactness"]


def test_python_random_known_answer():
    import oracle as orc
    # SURVEY.md Appendix A.1
    got = orc.py_random_stream(10, 3)
    assert got.tolist() == [0.5714025946899135, 0.4288890546751146, 0.5780913011344704]
    import random
    for seed in (0, 1, 41, 123456789, 2**32 + 5, 2**63 + 11):
        r = random.Random(seed)
        want = [r.random() for _ in range(1500)]
        assert orc.py_random_stream(seed, 1500).tolist() == want


@pytest.mark.parametrize("case", RMSA_CASES)
def test_rmsa_trace_bit_exact(case):
    z, meta = load_golden(case)
    topo = load_topology(meta["topology"])
    env = oracle_env_from_kwargs(topo, meta["env_kwargs"])
    n = meta["steps"]
    actions = None
    policy = meta["policy"]
    if policy == "random":
        actions = np.stack([z["act_path"], z["act_slot"]], axis=1).astype(np.int32)
        policy = "external"
    tr = env.run(policy, n, reset_on_done=meta["reset_on_done"], actions=actions)
    assert np.array_equal(tr["src"], z["src_id"])
    assert np.array_equal(tr["dst"], z["dst_id"])
    assert np.array_equal(tr["act_path"], z["act_path"])
    assert np.array_equal(tr["act_slot"], z["act_slot"])
    for f in INT_FIELDS:
        assert np.array_equal(tr[f].astype(np.int64), z[f].astype(np.int64)), f
    # bit_rate_selection="continuous" (rmsa_env.py:95-101): the reference keeps no histograms and no per-rate info keys
    discrete = meta["env_kwargs"].get("bit_rate_selection", "discrete") == "discrete"
    for f in FLOAT_FIELDS:
        if f == "fairness" and not discrete:
            continue
        a, b = tr[f], z[f]
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, (f, bad[:5], a[bad[:5]], b[bad[:5]])
    # final state
    av = env.available_slots()
    assert np.array_equal(np.packbits(av, axis=1, bitorder="little"), z["final_available_slots"])
    assert zlib.crc32(np.packbits(av, axis=1, bitorder="little").tobytes()) == int(z["occ_crc"][-1])
    ls = env.link_stats()
    assert np.array_equal(ls["utilization"], z["final_link_utilization"])
    assert np.array_equal(ls["external_fragmentation"], z["final_link_external_fragmentation"])
    assert np.array_equal(ls["compactness"], z["final_link_compactness"])
    assert np.array_equal(ls["last_update"], z["final_link_last_update"])
    if discrete:
        h = env.bit_rate_hist()
        assert np.array_equal(h["requested"], z["final_bit_rate_requested_hist"])
        assert np.array_equal(h["provisioned"], z["final_bit_rate_provisioned_hist"])
        assert np.array_equal(h["episode_requested"], z["final_episode_bit_rate_requested_hist"])
        assert np.array_equal(h["episode_provisioned"], z["final_episode_bit_rate_provisioned_hist"])
    r = env.request()
    assert [r.src, r.dst, r.bit_rate, r.service_id] == z["pending"].tolist()
    assert [r.arrival_time, r.holding_time] == z["pending_times"].tolist()


def test_survey_known_answers(nsfnet):
    """RMSA NSFNET S=320 load 50 seed 10 SAP-FF, continuous run."""
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=10)
    env = oracle_env_from_kwargs(nsfnet, kw)
    tr = env.run("sap_ff", 5000)
    assert (tr["act_path"][:6].tolist(), tr["act_slot"][:6].tolist()) == ([0] * 6, [0, 0, 0, 41, 41, 0])
    for n, proc, acc, req, prov, free in ((100, 101, 87, 69750, 55450, 3613), (500, 501, 366, 355800, 232800, 4215),
                                          (1000, 1001, 756, 701600, 482550, 4162),
                                          (5000, 5001, 3839, 3449750, 2404150, 4277)):
        i = n - 1
        assert (tr["services_processed"][i], tr["services_accepted"][i], tr["bit_rate_requested"][i],
                tr["bit_rate_provisioned"][i], tr["free_total"][i]) == (proc, acc, req, prov, free)
    # evaluate_heuristic-style: reset() between two episodes of 999 steps
    env = oracle_env_from_kwargs(nsfnet, kw)
    tr = env.run("sap_ff", 1998, reset_on_done=True)
    assert tr["done"].sum() == 2 and tr["done"][998] == 1 and tr["done"][1997] == 1
    assert tr["services_processed"][-1] == 1999 and tr["services_accepted"][-1] == 1494
    assert tr["episode_services_processed"][-1] - tr["episode_services_accepted"][-1] == 261


def test_oracle_asan(nsfnet):
    """Sanitizer run of the CPU oracle (ASan + UBSan) in a child process."""
    import subprocess
    import sys
    import oracle as orc
    path = orc.build(asan=True)
    code = (
        "import sys; sys.path[:0]=[%r,%r,%r]\n"
        "from conftest import *\n"
        "t=load_topology('nsfnet_chen_5-paths_6-modulations')\n"
        "e=oracle_env_from_kwargs(t, dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25,"
        " episode_length=100, seed=3), asan=True)\n"
        "tr=e.run('llp_ff', 600, reset_on_done=True); e.reset(False); tr=e.run('sap_ff', 300); e.close(); print('ok')\n"
    ) % (os.path.dirname(os.path.dirname(GOLDEN)), os.path.dirname(GOLDEN), os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), "oracle"))
    import glob as g
    asan = g.glob("/usr/lib/gcc/x86_64-linux-gnu/*/libasan.so") + g.glob("/usr/lib/x86_64-linux-gnu/libasan.so*")
    if not asan:
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=asan[0], ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]


def test_oracle_seed_between_steps_vs_reference():
    """OpticalNetworkEnv.seed called between two steps (optical_network_env.py:266-271): the fixture is the reference's own run
    (make_golden.py::gen_seed: seed(77) before step 40, seed() -- the default 41 -- before step 80); the oracle's seed() must
    give the same requests, times and decisions bit for bit."""
    z, meta = load_golden("seed_rmsa_nsfnet_s10")
    topo = load_topology(meta["topology"])
    o = oracle_env_from_kwargs(topo, meta["env_kwargs"])
    parts = []
    for t0, t1, sd in ((0, 40, "keep"), (40, 80, 77), (80, 120, None)):
        if sd != "keep":
            o.seed(sd)
        parts.append(o.run("sap_ff", t1 - t0))
    for f_ in ("src", "dst", "bit_rate", "arrival", "holding", "act_path", "act_slot", "accepted"):
        got = np.concatenate([p[f_] for p in parts])
        assert np.array_equal(got, z[f_]), f_
    o.close()
