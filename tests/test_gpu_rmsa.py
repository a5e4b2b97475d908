"""GPU parity tests proper: the HIP path (through the C ABI, liborlg.so) against the CPU oracle on the
same seeds, and against the golden traces recorded from the reference.

Bars
  * decisions (path, slot), accepted flags, every integer counter, histogram and the final
    occupancy bitmap: bit-exact against the golden traces of the reference AND against the oracle;
  * floats (arrival / holding times, compactness, time-weighted link and graph statistics):
    bit-exact against the oracle when the oracle's expovariate uses the library's host build of the
    device log (orlg_host_log, same IEEE operation sequence); against the reference's own floats
    (platform libm log) within rtol 1e-12 -- the two logs agree to 1 ulp (tests/test_host_logic.py).
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, load_topology, oracle_env_from_kwargs

pytestmark = pytest.mark.gpu

RMSA_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "rmsa_*.npz")))
OUTS = ("act_path", "act_slot", "accepted", "done", "reward", "request", "arrival", "holding",
        "network_compactness", "network_compactness_difference")


STEP_KERNEL = "auto"


@pytest.fixture(autouse=True, params=["wave", "group"])
def step_kernel(request):
    """Every test of this module runs against both step kernels (include/orlg.h ORLG_KERNEL_*): one wavefront per
    environment, and four environments per wavefront (which serves the first-fit policies and external actions and hands
    every other policy to the first)."""
    global STEP_KERNEL
    STEP_KERNEL = request.param
    yield request.param
    STEP_KERNEL = "auto"


def make_batched(topo, kw, batch, **extra):
    from optical_rl_gym_amd import BatchedRMSAEnv
    kw = dict(kw)
    kw.pop("allow_rejection", None)
    kw.pop("reset", None)
    extra.setdefault("step_kernel", STEP_KERNEL)
    return BatchedRMSAEnv(topo, batch, **kw, **extra)


@pytest.fixture()
def device_log_in_oracle():
    """Drive the oracle's expovariate with the library's host build of the device log."""
    import oracle as orc
    from optical_rl_gym_amd import _lib
    L = _lib.load()
    fn = C.cast(L.orlg_host_log, C.c_void_p).value
    orc.set_log_fn(fn)
    yield
    orc.set_log_fn(None)


def compare_with_oracle(env, topo, kw, policy, n, batch, reset_on_done, actions=None, seeds=None, j=1,
                        reward_mode=0):
    """Run the device for n steps and the oracle per env; everything must be bit-identical."""
    dev_policy = policy
    tr = env.run(dev_policy, n, outputs=OUTS, auto_reset=reset_on_done) if actions is None else None
    if actions is not None:
        # external actions are one launch per step
        cols = {k: [] for k in OUTS}
        for t in range(n):
            r = env.run(policy, 1, actions=actions[t], outputs=OUTS, auto_reset=reset_on_done)
            for k in OUTS:
                cols[k].append(r[k][0])
        tr = {k: np.stack(v) for k, v in cols.items()}
    cnt = env.counters()
    occ = env.available_slots()
    ls = env.link_stats()
    gs = env.graph_stats()
    hist = env.bit_rate_hist()
    req = env.requests()
    now = env.current_time()
    nrun = env.num_running()
    base_seed = kw.get("seed", 41)
    for i in range(batch):
        o = oracle_env_from_kwargs(topo, kw, seed=(base_seed + i) if seeds is None else int(seeds[i]), j=j,
                                   reward_mode=reward_mode)
        a = None
        if actions is not None:
            a = np.ascontiguousarray(actions[:, i])
        ot = o.run(policy, n, reset_on_done=reset_on_done, actions=a)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["act_slot"][:, i], ot["act_slot"]), i
        assert np.array_equal(tr["accepted"][:, i], ot["accepted"]), i
        assert np.array_equal(tr["done"][:, i], ot["done"]), i
        assert np.array_equal(tr["reward"][:, i], ot["reward"]), i
        assert np.array_equal(tr["request"][:, i, 0], ot["service_id"]), i
        assert np.array_equal(tr["request"][:, i, 1], ot["src"]), i
        assert np.array_equal(tr["request"][:, i, 2], ot["dst"]), i
        assert np.array_equal(tr["request"][:, i, 3], ot["bit_rate"]), i
        for f, g in (("arrival", "arrival"), ("holding", "holding"), ("network_compactness", "network_compactness"),
                     ("network_compactness_difference", "network_compactness_difference")):
            bad = np.nonzero(tr[f][:, i] != ot[g])[0]
            assert bad.size == 0, (f, i, bad[:4], tr[f][bad[:4], i], ot[g][bad[:4]])
        oc = o.counters()
        for name in oc:
            assert cnt[name][i] == oc[name], (name, i)
        assert np.array_equal(occ[i], o.available_slots()), i
        ols = o.link_stats()
        for name in ols:
            assert np.array_equal(ls[name][i], ols[name]), (name, i, ls[name][i], ols[name])
        ogs = o.graph_stats()
        for name in ogs:
            assert gs[name][i] == ogs[name], (name, i)
        oh = o.bit_rate_hist()
        for name in oh:
            assert np.array_equal(hist[name][i], oh[name]), (name, i)
        r = o.request()
        assert (req[i]["service_id"], req[i]["src"], req[i]["dst"], req[i]["bit_rate"]) == \
            (r.service_id, r.src, r.dst, r.bit_rate)
        assert (req[i]["arrival_time"], req[i]["holding_time"]) == (r.arrival_time, r.holding_time)
        assert now[i] == o.current_time()
        assert nrun[i] == o.num_running()
        o.close()
    return tr


@pytest.mark.parametrize("case", RMSA_CASES)
def test_golden_case_vs_reference_and_oracle(case, device_log_in_oracle):
    z, meta = load_golden(case)
    topo = load_topology(meta["topology"])
    kw = meta["env_kwargs"]
    n, batch = meta["steps"], 6
    policy = meta["policy"]
    actions = None
    if policy == "random":
        # env 0 replays the recorded random actions, the other envs get fresh ones
        rng = np.random.default_rng(123)
        K, S = topo.k_paths, kw["num_spectrum_resources"]
        actions = np.stack([rng.integers(0, K + 1, (n, batch)), rng.integers(0, S + 1, (n, batch))], axis=-1).astype(np.int32)
        actions[:, 0, 0] = z["act_path"]
        actions[:, 0, 1] = z["act_slot"]
        policy = "external"
        n = min(n, 400)
        actions = actions[:n]
    env = make_batched(topo, kw, batch)
    tr = compare_with_oracle(env, topo, kw, policy, n, batch, meta["reset_on_done"], actions=actions)
    # every policy runs on the kernel that was asked for (load balancing too)
    assert env.last_kernel().startswith("orlg_rmsa_group_kernel" if STEP_KERNEL == "group" else "orlg_rmsa_kernel"), env.last_kernel()
    # env 0 == the reference's own trace: decisions and integer state exactly, floats to rtol 1e-12
    assert np.array_equal(tr["act_path"][:, 0], z["act_path"][:n])
    assert np.array_equal(tr["act_slot"][:, 0], z["act_slot"][:n])
    assert np.array_equal(tr["accepted"][:, 0], z["accepted"][:n])
    assert np.array_equal(tr["done"][:, 0], z["done"][:n])
    assert np.array_equal(tr["request"][:, 0, 0], z["service_id"][:n])
    assert np.array_equal(tr["request"][:, 0, 1], z["src_id"][:n])
    assert np.array_equal(tr["request"][:, 0, 2], z["dst_id"][:n])
    assert np.array_equal(tr["request"][:, 0, 3], z["bit_rate"][:n])
    np.testing.assert_allclose(tr["arrival"][:, 0], z["arrival"][:n], rtol=1e-12, atol=0)
    np.testing.assert_allclose(tr["holding"][:, 0], z["holding"][:n], rtol=1e-12, atol=0)
    np.testing.assert_allclose(tr["network_compactness"][:, 0], z["network_compactness"][:n], rtol=1e-12, atol=0)
    if n == meta["steps"]:
        cnt = env.counters()
        for name in ("services_processed", "services_accepted", "bit_rate_requested", "bit_rate_provisioned",
                     "episode_services_processed", "episode_services_accepted"):
            assert cnt[name][0] == z[name][-1], name
        occ = env.available_slots()[0]
        assert np.array_equal(np.packbits(occ, axis=1, bitorder="little"), z["final_available_slots"])
        ls = env.link_stats()
        np.testing.assert_allclose(ls["utilization"][0], z["final_link_utilization"], rtol=1e-11)
        np.testing.assert_allclose(ls["external_fragmentation"][0], z["final_link_external_fragmentation"], rtol=1e-11)
        np.testing.assert_allclose(ls["compactness"][0], z["final_link_compactness"], rtol=1e-11)
        if "final_bit_rate_requested_hist" in z.files:   # (bit_rate_selection="continuous": the reference keeps no histograms)
            h = env.bit_rate_hist()
            assert np.array_equal(h["requested"][0], z["final_bit_rate_requested_hist"])
            assert np.array_equal(h["provisioned"][0], z["final_bit_rate_provisioned_hist"])
    env.close()


@pytest.mark.parametrize("stats_level", ["counters", "network"])
def test_lighter_stats_levels_keep_decisions(stats_level, nsfnet, device_log_in_oracle):
    """The cheaper statistics levels must not change decisions, counters or occupancy."""
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=10)
    z, _ = load_golden("rmsa_nsfnet_s10_sapff")
    env = make_batched(nsfnet, kw, 4, stats_level=stats_level)
    tr = env.run("sap_ff", 1500, outputs=("act_path", "act_slot", "accepted", "network_compactness"))
    assert np.array_equal(tr["act_path"][:, 0], z["act_path"][:1500])
    assert np.array_equal(tr["act_slot"][:, 0], z["act_slot"][:1500])
    assert np.array_equal(tr["accepted"][:, 0], z["accepted"][:1500])
    if stats_level == "network":
        np.testing.assert_allclose(tr["network_compactness"][:, 0], z["network_compactness"][:1500], rtol=1e-12)
    assert env.counters()["services_accepted"][0] == z["services_accepted"][1499]
    env.close()


def test_chunked_launches_equal_one_launch(nsfnet):
    """State survives the HBM round trip: 1 x 600 steps == 600 x 1 step == 7 uneven chunks."""
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=100, seed=77)
    a = make_batched(nsfnet, kw, 5)
    b = make_batched(nsfnet, kw, 5)
    c = make_batched(nsfnet, kw, 5)
    ta = a.run("sap_ff", 600, outputs=("act_slot", "accepted", "arrival"), auto_reset=True)
    tb = [b.run("sap_ff", 1, outputs=("act_slot", "accepted", "arrival"), auto_reset=True) for _ in range(600)]
    chunks = [1, 63, 62, 124, 200, 149, 1]
    tc = [c.run("sap_ff", n, outputs=("act_slot", "accepted", "arrival"), auto_reset=True) for n in chunks]
    for k in ("act_slot", "accepted", "arrival"):
        assert np.array_equal(ta[k], np.concatenate([t[k] for t in tb]))
        assert np.array_equal(ta[k], np.concatenate([t[k] for t in tc]))
    for env in (b, c):
        assert np.array_equal(a.occupancy_words(), env.occupancy_words())
        ca, cb = a.counters(), env.counters()
        for name in ca:
            assert np.array_equal(ca[name], cb[name])
        la, lb = a.link_stats(), env.link_stats()
        for name in la:
            assert np.array_equal(la[name], lb[name])
        assert np.array_equal(a.episodes_done(), env.episodes_done())
    for env in (a, b, c):
        env.close()


def test_batch_4096_properties(nsfnet, device_log_in_oracle):
    """BASELINE config 2 size: size-independent properties + spot checks against the oracle."""
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=10)
    B, n = 4096, 500
    env = make_batched(nsfnet, kw, B)
    tr = env.run("sap_ff", n, outputs=("accepted", "act_path", "act_slot"))
    cnt = env.counters()
    assert np.all(cnt["services_processed"] == n + 1)
    assert np.array_equal(cnt["services_accepted"], tr["accepted"].sum(axis=0))
    assert np.all((tr["act_path"] == 5) == (tr["accepted"] == 0))  # SAP-FF only proposes feasible actions
    # conservation: used slot-hops in the bitmap == sum over running services, checked through the oracle for a sample
    occ = env.available_slots()
    for i in (0, 1, 63, 64, 1000, 4095):
        o = oracle_env_from_kwargs(nsfnet, kw, seed=10 + i)
        ot = o.run("sap_ff", n)
        assert np.array_equal(tr["act_slot"][:, i], ot["act_slot"]), i
        assert np.array_equal(occ[i], o.available_slots()), i
        o.close()
    red, _ = env.reduce_counters()
    assert red["services_processed"] == int(cnt["services_processed"].sum())
    assert red["services_accepted"] == int(cnt["services_accepted"].sum())
    assert red["num_envs"] == B
    env.close()


def test_work_queue_more_envs_than_resident_waves(nsfnet, device_log_in_oracle):
    """B = 20 000 > the 4096 waves a MI355X keeps resident: the long launch hands environments out through the ticket
    counter (several launches: the host-side ticket base must advance correctly), the short ones stride statically.
    Sampled environments, incl. the first and last of every kind, are bit-exact against the oracle."""
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=77)
    B = 20000
    env = make_batched(nsfnet, kw, B)
    env.run("sap_ff", 120)                                   # ticket mode
    for _ in range(3):
        env.run("sap_ff", 5)                                 # static striding
    tr = env.run("sap_ff", 100, outputs=("act_path", "act_slot", "accepted"))   # ticket mode again, with outputs
    cnt = env.counters()
    assert np.all(cnt["services_processed"] == 236)
    occ, now = env.available_slots(), env.current_time()
    for i in (0, 4095, 4096, 4097, 8191, 8192, 12345, 19999):
        o = oracle_env_from_kwargs(nsfnet, kw, seed=77 + i)
        o.run("sap_ff", 135, fields=[])
        ot = o.run("sap_ff", 100)
        assert np.array_equal(tr["act_slot"][:, i], ot["act_slot"]) and np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(occ[i], o.available_slots()) and now[i] == o.current_time(), i
        assert cnt["services_accepted"][i] == o.counters()["services_accepted"], i
        o.close()
    env.close()


def test_full_reset_keeps_rng_stream(nsfnet, device_log_in_oracle):
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=5)
    env = make_batched(nsfnet, kw, 3)
    env.run("sap_ff", 200)
    env.reset(only_episode_counters=False)
    tr = env.run("sap_ff", 150, outputs=("act_slot", "arrival"))
    for i in range(3):
        o = oracle_env_from_kwargs(nsfnet, kw, seed=5 + i)
        o.run("sap_ff", 200, fields=[])
        o.reset(False)
        ot = o.run("sap_ff", 150)
        assert np.array_equal(tr["act_slot"][:, i], ot["act_slot"])
        assert np.array_equal(tr["arrival"][:, i], ot["arrival"])
        o.close()
    env.close()


def test_path_masks_query(nsfnet):
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=10)
    env = make_batched(nsfnet, kw, 2)
    env.run("sap_ff", 300)
    o = oracle_env_from_kwargs(nsfnet, kw, seed=11)
    o.run("sap_ff", 300, fields=[])
    masks, ns = env.path_masks(1)
    bits = np.unpackbits(masks.view(np.uint8), axis=-1, bitorder="little")[:, :320]
    for idp in range(5):
        assert ns[idp] == o.number_slots(idp)
        for s in range(0, 320, 7):
            want = o.is_path_free(idp, s, int(ns[idp]))
            got = s + ns[idp] <= 320 and bool(bits[idp, s:s + ns[idp]].all())
            assert want == got
    env.close()
    o.close()


@pytest.mark.parametrize("batch", [9, 11])
def test_group_kernel_odd_word_count_partial_quad(batch, device_log_in_oracle):
    """JPN12: 17 links x 5 words = 85 words per bitmap (odd), and a batch whose last quad holds one / three environments: the
    quad-wide linear copies end on an 8-byte tail.  Long and one-step launches of the group kernel against the oracle."""
    topo = load_topology("jpn12_5-paths_6-modulations")
    kw = dict(num_spectrum_resources=320, load=40, mean_service_holding_time=25, episode_length=70, seed=55)
    env = make_batched(topo, kw, batch, step_kernel="group")
    outs = ("act_path", "act_slot", "accepted", "network_compactness")
    parts = [env.run("sap_ff", n, outputs=outs, auto_reset=True) for n in (90, 1, 1, 2, 150, 1, 55)]
    tr = {k: np.concatenate([q[k] for q in parts]) for k in outs}
    occ = env.occupancy_words()
    for i in (0, batch - 2, batch - 1):
        o = oracle_env_from_kwargs(topo, kw, seed=55 + i)
        ot = o.run("sap_ff", 300, reset_on_done=True)
        assert np.array_equal(tr["act_path"][:, i], ot["act_path"]), i
        assert np.array_equal(tr["act_slot"][:, i], ot["act_slot"]), i
        assert np.array_equal(tr["network_compactness"][:, i], ot["network_compactness"]), i
        o.close()
    assert occ.shape[0] == batch
    env.close()


def test_group_kernel_short_launches_keep_the_queue_in_hbm(nsfnet):
    """Launches of at most four steps run the group kernel's instantiation that leaves the release queue in HBM (half the LDS
    per environment): a batch that is not a multiple of four (the idle row must not touch the last environment's ring), stepped
    1 / 3 / 4 steps at a time across episode ends, against an uninterrupted run of the wave-per-environment kernel -- outputs,
    counters, link statistics and the saved state byte for byte."""
    kw = dict(num_spectrum_resources=320, load=80, mean_service_holding_time=25, episode_length=60, seed=33)
    B = 9
    outs = ("act_path", "act_slot", "accepted", "arrival", "network_compactness", "done")
    ref = make_batched(nsfnet, kw, B, step_kernel="wave")
    env = make_batched(nsfnet, kw, B, step_kernel="group")
    t_ref = ref.run("sap_ff", 400, outputs=outs, auto_reset=True)
    parts, left, sizes = [], 400, (1, 3, 4, 1, 2)
    i = 0
    while left:
        n = min(sizes[i % len(sizes)], left)
        parts.append(env.run("sap_ff", n, outputs=outs, auto_reset=True))
        assert "<5,2,true>" in env.last_kernel(), env.last_kernel()
        left -= n
        i += 1
    for k in outs:
        assert np.array_equal(t_ref[k], np.concatenate([q[k] for q in parts])), k
    ca, cb = ref.counters(), env.counters()
    for name in ca:
        assert np.array_equal(ca[name], cb[name]), name
    la, lb = ref.link_stats(), env.link_stats()
    for name in la:
        assert np.array_equal(la[name], lb[name]), name
    assert np.array_equal(ref.save_state(), env.save_state())
    ref.close()
    env.close()


def test_group_kernel_ring_head_laps_the_queue(nsfnet):
    """A long launch pops several times the ring's capacity: the head goes round the ring, and the slots it emptied beyond
    (q_head - q_head0) % Q + q_n must reach HBM as (+inf, 0) too (the group kernel writes back the part of the ring it touched,
    pops + live entries, not the head's distance modulo Q).  Low load, the smallest queue, 1000-step launches: the saved state
    equals the wave kernel's byte for byte after every launch, and a state handed from the group kernel to the wave kernel
    whose queue drains to empty releases nothing that is not there."""
    kw = dict(num_spectrum_resources=320, load=6, mean_service_holding_time=10, episode_length=500, seed=5, queue_capacity=64)
    B = 37
    outs = ("act_path", "act_slot", "accepted", "arrival")
    a = make_batched(nsfnet, kw, B, step_kernel="wave")
    b = make_batched(nsfnet, kw, B, step_kernel="group")
    for n in (1000, 1000, 333, 1000):
        ta = a.run("sap_ff", n, outputs=outs, auto_reset=True)
        tb = b.run("sap_ff", n, outputs=outs, auto_reset=True)
        assert "group_kernel<5,2,false,true>" in b.last_kernel() or (n < 16 and "group_kernel<5,2>" in b.last_kernel()), b.last_kernel()   # the ring in LDS
        for k in outs:
            assert np.array_equal(ta[k], tb[k]), (n, k)
        assert np.array_equal(a.save_state(), b.save_state()), n
    # group -> wave hand-over, then a stretch long enough for every queue to run empty at this load
    a.load_state(b.save_state())
    ta = a.run("sap_ff", 400, outputs=outs, auto_reset=True)
    tb = b.run("sap_ff", 400, outputs=outs, auto_reset=True)
    for k in outs:
        assert np.array_equal(ta[k], tb[k]), k
    ca, cb = a.counters(), b.counters()
    for name in ca:
        assert np.array_equal(ca[name], cb[name]), name
    assert np.array_equal(a.occupancy_words(), b.occupancy_words())
    assert np.array_equal(a.save_state(), b.save_state())
    a.close()
    b.close()


def test_kernels_continue_each_other(nsfnet):
    """One state format: a batch stepped by the wave-per-environment kernel is handed (save_state / load_state) to the
    four-environments-per-wave kernel and back; outputs and final state equal an uninterrupted run."""
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=150, seed=21)
    B = 7   # not a multiple of four: the last wave of the group kernel has an idle row
    outs = ("act_path", "act_slot", "accepted", "arrival", "network_compactness", "done")
    ref = make_batched(nsfnet, kw, B, step_kernel="wave")
    a = make_batched(nsfnet, kw, B, step_kernel="wave")
    b = make_batched(nsfnet, kw, B, step_kernel="group")
    t_ref = ref.run("sap_ff", 700, outputs=outs, auto_reset=True)
    parts = []
    cur, other = a, b
    for n in (130, 1, 200, 69, 300):
        parts.append(cur.run("sap_ff", n, outputs=outs, auto_reset=True))
        other.load_state(cur.save_state())
        cur, other = other, cur
    for k in outs:
        assert np.array_equal(t_ref[k], np.concatenate([p[k] for p in parts])), k
    assert np.array_equal(ref.occupancy_words(), cur.occupancy_words())
    ca, cb = ref.counters(), cur.counters()
    for name in ca:
        assert np.array_equal(ca[name], cb[name]), name
    la, lb = ref.link_stats(), cur.link_stats()
    for name in la:
        assert np.array_equal(la[name], lb[name]), name
    assert np.array_equal(ref.episodes_done(), cur.episodes_done())
    assert np.array_equal(ref.save_state(), cur.save_state())   # byte for byte: queue slots, RNG state, arrival ring, caches
    for env in (ref, a, b):
        env.close()


def test_random_configurations_cross_check(step_kernel):
    """tools/cross_check_kernels.py: random grids, slot counts, k, loads, policies, batch sizes and launch lengths; the two step
    kernels must leave byte-identical states after every launch (450 configurations / 41 000 launches were run this way at the
    end of round 1; the suite keeps a dozen)."""
    if step_kernel != "wave":
        pytest.skip("one run covers both kernels")
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "cross_check_kernels.py"), "--configs", "12", "--seed", "7"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["configs_checked"] == 12


def test_reseed_between_launches_vs_oracle(nsfnet, device_log_in_oracle, step_kernel):
    """orlg_reseed: a fresh generator for every environment between two launches -- pending requests kept, arrivals
    pre-generated from the old generator dropped -- against the oracle's reseed() (all five draws from the new generator: NOT
    the reference's seed(), whose bit-rate draw stays with the old generator: include/orlg.h, tests/test_oracle_golden.py)."""
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=10)
    B = 6
    outs = ("act_path", "act_slot", "accepted", "arrival", "holding", "request")
    env = make_batched(nsfnet, kw, B, step_kernel=step_kernel)
    parts = [env.run("sap_ff", 40, outputs=outs)]
    env.reseed(77)
    parts.append(env.run("sap_ff", 150, outputs=outs))
    env.reseed(seeds=np.arange(500, 500 + B, dtype=np.uint64) * 3)
    parts.append(env.run("sap_ff", 60, outputs=outs))
    for i in range(B):
        o = oracle_env_from_kwargs(nsfnet, kw, seed=10 + i)
        op = [o.run("sap_ff", 40)]
        o.reseed(77 + i)
        op.append(o.run("sap_ff", 150))
        o.reseed((500 + i) * 3)
        op.append(o.run("sap_ff", 60))
        for a, b in zip(parts, op):
            for f in ("act_path", "act_slot", "accepted", "arrival", "holding"):
                assert np.array_equal(a[f][:, i], b[f]), (f, i)
            assert np.array_equal(a["request"][:, i, 1], b["src"]) and np.array_equal(a["request"][:, i, 3], b["bit_rate"]), i
        assert np.array_equal(env.available_slots()[i], o.available_slots()), i
        o.close()
    env.close()


def test_deferred_link_statistics(nsfnet, device_log_in_oracle):
    """Long launches with full statistics run the step kernels' instantiations that LOG the links' float64 updates and works
    them off one link per lane (group_link_replay): the same operations on the same values.  Against the instantiation that
    does them in place (ORLG_NO_DEFER, and launches below 16 steps) -- saved state byte for byte after every launch, a replay
    forced by a full log (load 300 on 100 slots: a link collects 40 updates within a launch) -- and against the oracle's link
    statistics."""
    # (both step kernels defer: the wave-per-environment kernel replays with lane = link, link_replay<64>)
    kernel = "group" if STEP_KERNEL == "group" else "wave"
    plain, deferred = (("orlg_rmsa_group_kernel<2,2>", "orlg_rmsa_group_kernel<2,2,false,true>") if kernel == "group"
                       else ("orlg_rmsa_kernel_ff<2,2>", "orlg_rmsa_kernel_ff<2,2,true>"))
    kw = dict(num_spectrum_resources=100, load=300, mean_service_holding_time=25, episode_length=300, seed=17,
              bit_rates=[25, 50, 75, 100])
    B = 10
    outs = ("act_path", "act_slot", "accepted")
    a = make_batched(nsfnet, kw, B, step_kernel=kernel)
    os.environ["ORLG_NO_DEFER"] = "1"
    try:
        b = make_batched(nsfnet, kw, B, step_kernel=kernel)
        ref_runs = [b.run("sap_ff", n, outputs=outs, auto_reset=True) for n in (700, 16, 333)]
        assert b.last_kernel().startswith(plain), b.last_kernel()
        sb = b.save_state()
    finally:
        del os.environ["ORLG_NO_DEFER"]
    runs = [a.run("sap_ff", n, outputs=outs, auto_reset=True) for n in (700, 16, 333)]
    assert a.last_kernel().startswith(deferred), a.last_kernel()
    for x, y in zip(runs, ref_runs):
        for k in outs:
            assert np.array_equal(x[k], y[k]), k
    assert np.array_equal(a.save_state(), sb)
    ls = a.link_stats()
    for i in (0, 3, 9):
        o = oracle_env_from_kwargs(nsfnet, kw, seed=17 + i)
        o.run("sap_ff", 700 + 16 + 333, reset_on_done=True, fields=[])
        ols = o.link_stats()
        for name in ols:
            assert np.array_equal(ls[name][i], ols[name]), (name, i)
        o.close()
    a.close()
    b.close()
