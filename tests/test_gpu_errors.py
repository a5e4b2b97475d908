"""Error behaviour of the C ABI on the device: limits are refused with ORLG_ERR_INVALID and a message, an exhausted release
queue / channel_state list / work list is reported as ORLG_ERR_QUEUE_FULL by reduce_counters -- never silently dropped."""
import numpy as np
import pytest

from conftest import load_golden, load_phy_tables, load_topology
from test_gpu_phy import make_env
from test_gpu_rmsa import make_batched

pytestmark = pytest.mark.gpu


def test_rmsa_queue_overflow_is_reported(nsfnet):
    from optical_rl_gym_amd import OrlgError
    kw = dict(num_spectrum_resources=320, load=150, mean_service_holding_time=25, episode_length=1000, seed=1)
    env = make_batched(nsfnet, kw, 16, queue_capacity=64)   # well over 64 services in progress at this load
    env.run("sap_ff", 3000)
    with pytest.raises(OrlgError) as ei:
        env.reduce_counters()
    assert ei.value.code == -4 and "queue" in str(ei.value)
    env.close()


@pytest.mark.parametrize("kernel", ["wave", "group"])
def test_overflow_is_returned_by_step_and_synchronize(nsfnet, kernel):
    """A launch that loses a release makes the next call that waits for the stream fail: orlg_step with host outputs,
    orlg_synchronize -- not only orlg_reduce_counters; read-back still works and a full reset clears the condition."""
    from optical_rl_gym_amd import OrlgError
    kw = dict(num_spectrum_resources=320, load=150, mean_service_holding_time=25, episode_length=1000, seed=1)
    env = make_batched(nsfnet, kw, 16, queue_capacity=64, step_kernel=kernel)
    with pytest.raises(OrlgError) as ei:
        env.run("sap_ff", 3000, outputs=("accepted",))
    assert ei.value.code == -4
    env.run("sap_ff", 10)            # asynchronous launch: nothing to report yet
    with pytest.raises(OrlgError) as ei:
        env.synchronize()
    assert ei.value.code == -4
    assert env.num_running().max() >= 64      # state read-back is not blocked
    env.reset(only_episode_counters=False)
    env.run("sap_ff", 20, outputs=("accepted",))
    env.synchronize()
    env.close()


def test_load_state_recomputes_the_error_word(nsfnet):
    """The sticky ORLG_ERR_QUEUE_FULL describes the state the handle holds: loading a clean checkpoint -- the natural recovery
    -- clears it, loading the checkpoint of an overflowed batch brings it back (include/orlg.h, orlg_load_state)."""
    from optical_rl_gym_amd import OrlgError
    kw = dict(num_spectrum_resources=320, load=150, mean_service_holding_time=25, episode_length=1000, seed=1)
    env = make_batched(nsfnet, kw, 16, queue_capacity=64)
    env.run("sap_ff", 5)
    clean = env.save_state()
    with pytest.raises(OrlgError):
        env.run("sap_ff", 3000, outputs=("accepted",))
    bad = env.save_state()
    env.load_state(clean)
    env.synchronize()                                   # no longer reported
    r = env.run("sap_ff", 5, outputs=("accepted",))     # and the batch steps on from the checkpoint
    assert r["accepted"].shape == (5, 16)
    env.reduce_counters()
    env.load_state(bad)
    with pytest.raises(OrlgError) as ei:
        env.synchronize()
    assert ei.value.code == -4
    env.close()


def test_wrong_action_arrays_are_refused(nsfnet):
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=1)
    env = make_batched(nsfnet, kw, 8)
    with pytest.raises(ValueError):
        env.run("external", 1, actions=np.zeros(8, np.int32))
    with pytest.raises(ValueError):
        env.run("external", 1)
    with pytest.raises(TypeError):
        env.run("deeprmsa_external", 1, actions=np.zeros(8, np.float64))
    with pytest.raises(ValueError):
        env.run("sap_ff", 4, out={"accepted": np.zeros((3, 8), np.uint8)})
    with pytest.raises(TypeError):
        env.run("sap_ff", 4, out={"accepted": np.zeros((4, 8), np.int32)})
    r = env.run("external", 1, actions=np.zeros((8, 2), np.int64), outputs=("accepted",))   # integer arrays are converted
    assert r["accepted"].shape == (1, 8)
    env.close()


def test_group_kernel_overflow_and_lds_limit(nsfnet):
    """The four-environments-per-wave kernel reports a full release queue like the other one, and a shape whose four
    environments do not fit the LDS is refused when that kernel is demanded (AUTO falls back to the wave-per-environment kernel)."""
    from optical_rl_gym_amd import OrlgError
    kw = dict(num_spectrum_resources=320, load=150, mean_service_holding_time=25, episode_length=1000, seed=1)
    env = make_batched(nsfnet, kw, 16, queue_capacity=64, step_kernel="group")
    env.run("sap_ff", 3000)
    with pytest.raises(OrlgError) as ei:
        env.reduce_counters()
    assert ei.value.code == -4 and "queue" in str(ei.value)
    env.close()
    kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, episode_length=1000, seed=1)
    with pytest.raises(OrlgError) as ei:
        make_batched(nsfnet, kw, 8, queue_capacity=4096, step_kernel="group")
    assert ei.value.code == -1 and "LDS" in str(ei.value)
    env = make_batched(nsfnet, kw, 8, queue_capacity=4096, step_kernel="auto")
    env.run("sap_ff", 50)
    env.close()


def test_rmsa_limits_are_refused(nsfnet):
    from optical_rl_gym_amd import OrlgError
    for bad in (dict(num_spectrum_resources=513), dict(num_spectrum_resources=0), dict(load=0)):
        kw = dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, seed=1)
        kw.update(bad)
        with pytest.raises((OrlgError, ZeroDivisionError, ValueError, AssertionError)):
            make_batched(nsfnet, kw, 4).close()
    with pytest.raises(OrlgError) as ei:
        make_batched(nsfnet, dict(num_spectrum_resources=320, load=50, mean_service_holding_time=25, seed=1), 0)
    assert ei.value.code == -1


def test_phy_small_structures_overflow_is_reported():
    from optical_rl_gym_amd import OrlgError
    z, meta = load_golden("phy_us14_s10_sapff")
    topo, tables = load_topology(meta["topology"]), load_phy_tables(meta["tables"])
    env = make_env(topo, tables, dict(meta["env_kwargs"], load=3000), 4, queue_capacity=256)
    env.run("sapff", 1500, auto_reset=True)
    assert env.episode_stats()["queue_overflow"].max() != 0
    with pytest.raises(OrlgError) as ei:
        env.reduce_counters()
    assert ei.value.code == -4
    env.close()


def test_phy_level_zero_tables_are_refused():
    """A level-0 entry would sort FIRST in the reference (uint8 negation, SURVEY 8c caveat 3): refused, not reinterpreted."""
    from optical_rl_gym_amd import OrlgError
    z, meta = load_golden("phy_us14_s10_sapff")
    topo = load_topology(meta["topology"])
    pairs, mod, gsnr = load_phy_tables(meta["tables"])
    mod = mod.copy()
    mod[3, 100, 1] = 0
    with pytest.raises(OrlgError) as ei:
        make_env(topo, (pairs, mod, gsnr), meta["env_kwargs"], 2)
    assert ei.value.code == -1 and "modulation level" in str(ei.value)


def test_phy_wrong_action_arrays_are_refused():
    """External (path, channels) actions from host arrays: integers only and inside the ABI's int32 / int16, as
    BatchedRMSAEnv.run validates its actions -- a float or an out-of-range channel number is refused, not truncated or wrapped."""
    z, meta = load_golden("phy_us14_s10_sapff")
    topo, tables = load_topology(meta["topology"]), load_phy_tables(meta["tables"])
    env = make_env(topo, tables, meta["env_kwargs"], 4)
    ch = np.full((4, 14), -1, np.int64)
    with pytest.raises(TypeError):
        env.run("external", 1, act_path=np.zeros(4, np.float64), act_channels=ch)
    with pytest.raises(TypeError):
        env.run("external", 1, act_path=np.zeros(4, np.int32), act_channels=ch.astype(np.float32))
    big = ch.copy()
    big[0, 0] = 70000    # would wrap to 4464 as int16
    with pytest.raises(ValueError):
        env.run("external", 1, act_path=np.zeros(4, np.int32), act_channels=big)
    with pytest.raises(ValueError):
        env.run("external", 1, act_path=np.zeros(3, np.int32), act_channels=ch)
    r = env.run("external", 1, act_path=np.full(4, -2, np.int64), act_channels=ch, outputs=("accepted",))   # blocked: (-2, [])
    assert r["accepted"].shape == (1, 4) and not r["accepted"].any()
    env.close()
