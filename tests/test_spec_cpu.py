"""Spec front-end (mbfir.spec): known answers and round trips.  Host only."""
import math

import numpy as np
import pytest

import mbfir
from conftest import A_C13, D_C13, F100


def test_c13_bssfp_spec_matches_the_surveyed_values():
    """S-C13 at n=100 (evaluated there from rf_bandedge.m:133-161 and rf_ripple_GFA.m:210-224,287-292)."""
    f, a, d = mbfir.spec.spec_c13_bssfp(100)
    assert np.allclose(f, F100, atol=5e-7)
    assert np.allclose(a, A_C13, atol=5e-7)
    assert np.allclose(d, D_C13, atol=5e-9)


def test_fixed_duration_regime_scales_band_edges():
    f100, _, _ = mbfir.spec.spec_c13_bssfp(100)
    f512, a, d = mbfir.spec.spec_c13_bssfp(512)
    assert np.allclose(f512, f100 * 100.0 / 512.0, rtol=1e-12)
    assert np.allclose(a, A_C13, atol=5e-7) and np.allclose(d, D_C13, atol=5e-9)


@pytest.mark.parametrize("FA,rip", [(60, 0.01), (0, 0.005), (90, 0.02), (120, 0.03), (30, 0.001)])
def test_excitation_ripple_round_trip(FA, rip):
    """|Mxy| = 2 |B| sqrt(1 - |B|^2) at the ends of the returned |B| range sits on the ripple bounds."""
    lo, hi = mbfir.spec.rf_ripple_gfa(FA, rip, "ex")
    mxy = lambda b: 2 * b * math.sqrt(1 - b * b)
    nominal = math.sin(math.radians(FA))
    ends = sorted([mxy(lo), mxy(hi)]) if lo >= 0 else [mxy(hi)]
    assert abs(ends[0] - (nominal - rip if FA <= 90 and lo >= 0 else ends[0])) < 1e-12
    if nominal + rip < 1 and FA <= 90:
        assert abs(mxy(hi) - (nominal + rip)) < 1e-12
    if FA > 90:
        assert abs(mxy(hi) - (nominal - rip)) < 1e-12 and abs(mxy(lo) - (nominal + rip)) < 1e-12


@pytest.mark.parametrize("FA,rip", [(120, 0.05), (0, 0.001), (90, 0.05), (180, 0.02)])
def test_saturation_ripple_round_trip(FA, rip):
    """Mz = 1 - 2 |B|^2 at the ends of the range sits on cos(FA) -+ ripple (clipped to [-1, 1])."""
    lo, hi = mbfir.spec.rf_ripple_gfa(FA, rip, "sat")
    mz = lambda b: 1 - 2 * b * b
    c = math.cos(math.radians(FA))
    assert abs(mz(hi) - max(c - rip, -1.0)) < 1e-12
    if lo >= 0:
        assert abs(mz(lo) - min(c + rip, 1.0)) < 1e-12
    else:
        assert abs(lo + hi) < 1e-15          # range symmetric about 0 when the band may cross |B| = 0


def test_h1_dualband_spec():
    f, a, d = mbfir.spec.spec_h1_dualband(260)
    assert len(f) == 6 and np.all(np.diff(f) > 0)
    assert abs(f[0] + f[-1]) < 1e-15                       # shift_f = 1 centres the two high-flip-angle bands
    assert a[2] == a[3] == 0 and abs(d[1] - math.sqrt(0.0005)) < 1e-12
    f2, a2, d2 = mbfir.spec.spec_h1_dualband(512)          # dt snaps to a multiple of 4 us: 48 us, T = 24.576 ms
    assert np.allclose(a2, a) and np.allclose(d2, d)
    assert np.allclose(f2 / f, (0.048 / 0.1), rtol=1e-12)


def test_bandedge_from_dinf_and_error_paths():
    f = mbfir.spec.rf_bandedge(200, 0.02, [-1.0, 0.0, 1.5], None, [0, 90, 0], [0.005, 0.01, 0.005], "ex")
    assert len(f) == 6 and np.all(np.diff(f) > 0) and f[0] >= -1 and f[-1] <= 1
    with pytest.raises(ValueError):
        mbfir.spec.rf_bandedge(100, 0.04, [0.0, 0.05], [0.2, 0.2], [0, 60], [0.005, 0.01], "ex")      # overlapping bands
    with pytest.raises(ValueError):
        mbfir.spec.rf_bandedge(10, 0.4, [-2.0, 0.0], [0.1, 0.1], [0, 60], [0.005, 0.01], "ex")        # beyond Nyquist
    with pytest.raises(ValueError):
        mbfir.spec.rf_ripple_gfa(60, 0.01, "st")
    assert abs(mbfir.spec.dinf(0.01, 0.001) - 2.5351) < 2e-2   # Parks-McClellan D-infinity for (0.01, 0.001)


def test_spec_rand_is_the_survey_s_rand_workload():
    """S-RAND: k in 2..8 non-overlapping bands on [-1, 1], widths U(0.01, 0.1), gaps >= 8 / n, amplitudes 0 or
    U(0.2, 0.9) with at least one pass band, ripples U(0.002, 0.02); deterministic per seed (the bench's heterogeneous batch and the
    heterogeneous-unit tests draw from it)."""
    ks = set()
    for seed in range(64):
        f, a, d = mbfir.spec.spec_rand(512, seed)
        k = len(d)
        ks.add(k)
        assert 2 <= k <= 8 and len(f) == len(a) == 2 * k
        assert f[0] >= -1.0 and f[-1] <= 1.0 and np.all(np.diff(f) > 0)
        widths, gaps = f[1::2] - f[0::2], f[2::2] - f[1:-1:2]
        assert np.all(widths >= 0.01 - 1e-15) and np.all(widths <= 0.1 + 1e-15) and np.all(gaps >= 8.0 / 512 - 1e-12)
        assert np.all(a[0::2] == a[1::2]) and np.any(a > 0) and np.all((a == 0) | ((a >= 0.2) & (a <= 0.9)))
        assert np.all(d >= 0.002) and np.all(d <= 0.02)
        f2, a2, d2 = mbfir.spec.spec_rand(512, seed)
        assert np.array_equal(f, f2) and np.array_equal(a, a2) and np.array_equal(d, d2)
    assert ks == set(range(2, 9))
    with pytest.raises(ValueError):
        mbfir.spec.spec_rand(16, 0, kmin=8, kmax=8)          # eight bands with gaps of 8 / n do not fit at n = 16
