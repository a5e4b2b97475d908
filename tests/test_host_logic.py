"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/orlg.h declares, fails loudly without a GPU (there is NO CPU compute path), the host build of the
device log is within 1 ulp of libm, and the frozen-topology / selection tables are what the reference
builds."""
import ctypes as C
import math
import os
import random
import re

import numpy as np
import pytest

from conftest import ROOT, load_topology


def test_library_exports_every_declared_symbol():
    from optical_rl_gym_amd import _lib
    L = _lib.load()
    header = open(os.path.join(ROOT, "include", "orlg.h")).read()
    declared = sorted(set(re.findall(r"\b(orlg_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"liborlg.so does not export {name}"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    assert L.orlg_abi_version() == 3


def test_no_cpu_fallback_without_device(nsfnet):
    """On a box without a GPU creating an environment must fail with ORLG_ERR_NO_DEVICE."""
    from optical_rl_gym_amd import BatchedRMSAEnv, OrlgError, _lib
    if _lib.load().orlg_device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(OrlgError) as ei:
        BatchedRMSAEnv(nsfnet, 4, num_spectrum_resources=320, load=50, mean_service_holding_time=25)
    assert ei.value.code == -2


def test_host_log_within_one_ulp_of_libm():
    from optical_rl_gym_amd import _lib
    L = _lib.load()
    rng = random.Random(7)
    worst, differ, n = 0, 0, 200000
    for _ in range(n):
        x = 1.0 - rng.random()
        a, b = L.orlg_host_log(x), math.log(x)
        if a != b:
            differ += 1
            ia = np.float64(a).view(np.int64)
            ib = np.float64(b).view(np.int64)
            worst = max(worst, abs(int(ia) - int(ib)))
    assert worst <= 1
    assert differ / n < 0.15
    for x, want in ((1.0, 0.0), (0.5, math.log(0.5)), (2.0 ** -53, math.log(2.0 ** -53))):
        assert abs(L.orlg_host_log(x) - want) <= abs(want) * 2.3e-16


def test_decisions_do_not_depend_on_which_log(nsfnet):
    """The oracle driven by libm's log (= the reference) and by the device's log makes identical
    decisions on the golden configurations; times agree to rtol 1e-13."""
    import oracle as orc
    from conftest import oracle_env_from_kwargs
    from optical_rl_gym_amd import _lib
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=10)
    a = oracle_env_from_kwargs(nsfnet, kw).run("sap_ff", 3000)
    orc.set_log_fn(C.cast(_lib.load().orlg_host_log, C.c_void_p).value)
    try:
        b = oracle_env_from_kwargs(nsfnet, kw).run("sap_ff", 3000)
    finally:
        orc.set_log_fn(None)
    for f in ("act_path", "act_slot", "accepted", "services_accepted", "bit_rate_provisioned", "src", "dst", "bit_rate"):
        assert np.array_equal(a[f], b[f]), f
    np.testing.assert_allclose(a["arrival"], b["arrival"], rtol=1e-13)
    assert not np.array_equal(a["arrival"], b["arrival"])  # they really are different logs


def test_frozen_topology_tables(nsfnet):
    t = nsfnet
    assert (t.num_nodes, t.num_links, t.k_paths, t.num_paths) == (14, 22, 5, 455)
    assert t.path_hops.max() == 9 and abs(t.path_hops.mean() - 4.07) < 0.01
    assert np.bincount(t.path_se, minlength=7)[1:].tolist() == [744 // 2, 118 // 2, 36 // 2, 10 // 2, 2 // 2, 0]
    # ksp[a,b] and ksp[b,a] are the same list object (create_topology.py:136-137)
    assert t.ksp["1", "5"] is t.ksp["5", "1"]
    rec = t.packed_path_records()
    g = t.pair_path_base[0 * 14 + 4]
    assert rec[g, 0] == t.path_hops[g] and rec[g, 1] == t.path_se[g]
    v = t.view()
    assert v.number_of_nodes() == 14 and v.number_of_edges() == 22
    p = t.ksp["1", "5"][0]
    assert [v[p.node_list[i]][p.node_list[i + 1]]["index"] for i in range(p.hops)] == \
        t.path_links[t.path_link_off[g]:t.path_link_off[g + 1]].tolist()
    # JSON round trip
    from optical_rl_gym_amd import FrozenTopology
    t2 = FrozenTopology.from_json(t.to_json())
    assert np.array_equal(t2.path_links, t.path_links) and np.array_equal(t2.pair_path_base, t.pair_path_base)


def test_selection_tables_match_cpython_choices():
    from optical_rl_gym_amd import selection_tables
    probs = np.array([0.1, 0.2, 0.05, 0.25, 0.4])
    _, src_cum, dst_cum, br_cum = selection_tables(probs, [0.5, 0.25, 0.25], 5, [10, 20, 30])
    r1, r2 = random.Random(3), random.Random(3)
    import bisect
    for _ in range(2000):
        want = r1.choices(range(5), weights=probs)[0]
        got = bisect.bisect(list(src_cum), r2.random() * (src_cum[-1] + 0.0), 0, 4)
        assert want == got
        p = probs.copy(); p[want] = 0; p = p / np.sum(p)
        want_d = r1.choices(range(5), weights=p)[0]
        got_d = bisect.bisect(list(dst_cum[want]), r2.random() * (dst_cum[want][-1] + 0.0), 0, 4)
        assert want_d == got_d and got_d != want


def test_registry_ids_resolve():
    """optical_rl_gym/__init__.py:3-31: the ids of the hot-path environments resolve to the single-env views."""
    import optical_rl_gym_amd as pkg
    assert pkg.env_class("RMSA-v0") is pkg.RMSAEnv
    assert pkg.env_class("DeepRMSA-v0") is pkg.DeepRMSAEnv
    assert pkg.env_class("PhyRMSA-v0") is pkg.PhyRMSAEnv
    with pytest.raises(KeyError):
        pkg.env_class("RWA-v0")   # out of scope
    assert pkg.register_with_gym() is None or hasattr(pkg.register_with_gym(), "make")


def test_caller_buffers_are_validated():
    """Arrays the C ABI reads or writes through raw pointers are checked for shape, dtype and layout first (a [B] array
    for a [B, 2] action would be overrun, an int64 tensor reinterpreted)."""
    from optical_rl_gym_amd.batched import _check_buffer
    ok = np.zeros((3, 8), np.float64)
    assert _check_buffer("out['reward']", ok, (3, 8), "float64") is ok
    with pytest.raises(ValueError):
        _check_buffer("actions", np.zeros(8, np.int32), (8, 2), np.int32)
    with pytest.raises(TypeError):
        _check_buffer("actions", np.zeros((8, 2), np.int64), (8, 2), np.int32)
    with pytest.raises(ValueError):
        _check_buffer("out['done']", np.zeros((8, 6), np.uint8)[:, ::2], (8, 3), np.uint8)
    torch = pytest.importorskip("torch")
    t = torch.zeros((8, 2), dtype=torch.int32)
    assert _check_buffer("actions", t, (8, 2), np.int32) is t
    with pytest.raises(TypeError):
        _check_buffer("actions", torch.zeros((8, 2), dtype=torch.int64), (8, 2), np.int32)
    with pytest.raises(ValueError):
        _check_buffer("actions", torch.zeros((2, 8), dtype=torch.int32).t(), (8, 2), np.int32)
