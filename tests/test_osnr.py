"""GN-model GSNR routine: oracle vs the golden grid (CPU) and device vs oracle / golden (GPU), tolerance 1e-6
relative as north_star states (observed ~1e-15).  Parity unpinned by the reference itself (no caller / no test)."""
import numpy as np
import pytest

from conftest import GOLDEN
import os


def grid():
    return np.load(os.path.join(GOLDEN, "osnr_grid.npz"), allow_pickle=False)


def test_oracle_matches_golden_grid():
    import oracle as orc
    z = grid()
    got = orc.gn_osnr(z)
    np.testing.assert_allclose(got, z["gsnr_db"], rtol=1e-12)


def test_stale_phi_quirk_is_reproduced():
    """Putting the current service itself into a link's running list changes the result (the stale phi is added)."""
    import oracle as orc
    att = 0.2 / (2 * 10 * np.log10(np.exp(1)) * 1e3)
    base = dict(check_link_off=[0, 1], link_span_off=[0, 2], bandwidth=[50e9], center_frequency=[193.1e12],
                launch_power=[1e-3], span_length_km=[75.0, 75.0], span_attenuation=[att, att],
                span_noise_figure=[10 ** 0.55] * 2)
    without = dict(base, link_svc_off=[0, 1], svc_bandwidth=[50e9], svc_center_frequency=[193.2e12], svc_se=[2], svc_is_self=[0])
    with_self_after = dict(base, link_svc_off=[0, 2], svc_bandwidth=[50e9, 50e9], svc_center_frequency=[193.2e12, 193.1e12],
                           svc_se=[2, 1], svc_is_self=[0, 1])
    a, b = orc.gn_osnr(without)[0], orc.gn_osnr(with_self_after)[0]
    assert a != b and abs(a - b) < 1.0


@pytest.mark.gpu
def test_device_matches_oracle_and_golden():
    import oracle as orc
    from optical_rl_gym_amd import gn_osnr, modulation_level_from_gsnr
    z = grid()
    got = gn_osnr(z)
    np.testing.assert_allclose(got, z["gsnr_db"], rtol=1e-6)      # the stated tolerance
    np.testing.assert_allclose(got, z["gsnr_db"], rtol=1e-12)     # what is actually achieved
    np.testing.assert_allclose(got, orc.gn_osnr(z), rtol=1e-12)
    lv = modulation_level_from_gsnr(got)
    assert lv.min() >= 0 and lv.max() <= 6
    # empty batch and a check whose link lists are empty
    assert gn_osnr({k: z[k][:0] if k not in ("check_link_off", "link_span_off", "link_svc_off") else np.zeros(1, np.int32)
                    for k in z.files if k != "gsnr_db"}).shape == (0,)
