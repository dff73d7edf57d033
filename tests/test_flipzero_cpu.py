"""fir_flip_zero mirror (host only): magnitude response kept, peak not increased, the reference's enumeration order."""
import numpy as np
import pytest

import mbfir
from mbfir.flipzero import _masks


def minphase_like(n, seed):
    """A filter with zeros inside / on / outside the unit circle, like a minimum-phase beta polynomial with a stop band."""
    rng = np.random.default_rng(seed)
    zi = 0.7 * np.exp(1j * rng.uniform(-0.6, 0.6, n // 3))                  # pass-band zeros (inside)
    zo = np.exp(1j * rng.uniform(1.0, 2 * np.pi - 1.0, n - 1 - n // 3))       # stop-band zeros on the circle
    return np.poly(np.concatenate([zi, zo])) * 0.01


def test_masks_follow_combination_2power():
    # combination_2power(2) = [1 1 0 0; 1 0 1 0] (fir_flip_zero.m:153-160)
    assert _masks(2, None).tolist() == [[1, 1, 0, 0], [1, 0, 1, 0]]
    m = _masks(3, None)
    assert m.shape == (3, 8) and m[:, 0].tolist() == [1, 1, 1] and m[:, -1].tolist() == [0, 0, 0] and m[:, 1].tolist() == [1, 1, 0]


@pytest.mark.parametrize("n,seed", [(16, 0), (25, 1), (31, 2)])
def test_flip_keeps_the_magnitude_response_and_lowers_the_peak(n, seed):
    h = minphase_like(n, seed)
    hn, info = mbfir.fir_flip_zero(h, return_info=True)
    assert len(hn) == n and info["n_passband_zeros"] == n // 3 and info["candidates"] == 2 ** (n // 3)
    assert info["peak_after"] <= info["peak_before"] * (1 + 1e-12)
    H = np.abs(np.fft.fft(h, 1024))
    Hn = np.abs(np.fft.fft(hn, 1024))
    # reflecting a zero scales |H| by a constant; the DC rescaling (:83) restores it exactly when H(0) != 0
    assert np.max(np.abs(Hn - H)) <= 1e-8 * np.max(H)
    assert abs(np.sum(hn) - np.sum(h)) <= 1e-12 * abs(np.sum(h))


def test_all_minimum_phase_input_finds_a_lower_peak():
    h = minphase_like(28, 7)                      # every pass-band zero inside: the energy sits in the first taps
    hn = mbfir.fir_flip_zero(h)
    assert np.max(np.abs(hn)) < 0.9 * np.max(np.abs(h))


def test_more_than_twelve_zeros_samples_4096_candidates_reproducibly():
    h = minphase_like(46, 3)                      # 15 pass-band zeros
    h1, i1 = mbfir.fir_flip_zero(h, seed=11, return_info=True)
    h2, i2 = mbfir.fir_flip_zero(h, seed=11, return_info=True)
    assert i1["candidates"] == 4096 and np.array_equal(h1, h2)
    assert i1["peak_after"] <= i1["peak_before"]
    assert _masks(25, np.random.default_rng(0)).shape == (25, 4096)          # Monte-Carlo masks beyond 19 zeros


def test_no_passband_zero_returns_the_input():
    h = np.poly(np.exp(1j * np.linspace(0.5, 5.5, 9)))
    assert np.allclose(mbfir.fir_flip_zero(h), h)
