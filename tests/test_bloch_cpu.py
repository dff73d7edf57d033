"""Forward SLR / Bloch oracle (oracle/bloch.py): the identities that pin it, and the end of the chain
spec -> designer -> inverse SLR -> simulated profile on the CPU oracle alone."""
import json
import os
import warnings

import numpy as np
import pytest

import mbfir
from oracle import bloch, designers, slr

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "slr_golden.json")))


def cx(d):
    return np.array(d["re"]) + 1j * np.array(d["im"])


@pytest.mark.parametrize("name", ["ap_c13_64_p071", "sinc127_p026", "lin_cplx31_p071", "qp_modelA48_p026", "rand200_p071"])
def test_hard_pulse_simulation_inverts_the_inverse_slr(name):
    """b -> b2rf (pinned by the reference's C) -> hard-pulse simulation: |beta(x)| = |B(w)|, w = -2 pi x / n, and
    alpha is the minimum-phase partner (|a|^2 + |b|^2 = 1) -- the SLR identity, to rounding."""
    b = cx(GOLD[name]["b"])
    n = len(b)
    rf = slr.b2rf(b)
    x = np.linspace(-n / 2, n / 2, 257)[:-1]
    a1, b1 = bloch.hard_pulse_ab(rf, x)
    w = -2 * np.pi * x / n
    B = (b[None, :] * np.exp(1j * w[:, None] * np.arange(n)[None, :])).sum(1)
    assert np.max(np.abs(np.abs(b1) - np.abs(B))) < 1e-10
    assert np.max(np.abs(np.abs(a1) ** 2 + np.abs(b1) ** 2 - 1)) < 1e-12


def test_abrm_is_unitary_and_close_to_the_hard_pulse_model():
    b = cx(GOLD["sinc127_p026"]["b"])
    rf = slr.b2rf(b)
    x = np.linspace(-20, 20, 101)
    a0, b0 = bloch.abrm(rf, x)
    a1, b1 = bloch.hard_pulse_ab(rf, -x)                    # abrm's om = +x g is the hard-pulse model's z = e^{-i..} axis mirrored
    assert np.max(np.abs(np.abs(a0) ** 2 + np.abs(b0) ** 2 - 1)) < 1e-12
    assert np.max(np.abs(np.abs(b0) - np.abs(b1))) < 2e-3  # joint rotation vs split step: O(rf * om) per sample
    a2, b2 = bloch.abr(rf, x)
    assert np.array_equal(b2, -np.conj(b0)) and np.array_equal(a2, a0)


def test_abrm_on_resonance_hard_pulse_train():
    """x = 0: the rotations commute, the flip angle is sum(rf) (abrm.m header: 'rf scaled so that sum(rf) = flip angle')."""
    rf = np.full(40, (np.pi / 2) / 40)
    a, b = bloch.abrm(rf, [0.0])
    assert abs(a[0] - np.cos(np.pi / 4)) < 1e-14 and abs(b[0] - (-1j * np.sin(np.pi / 4))) < 1e-14
    assert abs(abs(bloch.mxy_excitation(a, b)[0]) - 1.0) < 1e-14 and abs(bloch.mz_inversion(b)[0]) < 1e-14


def test_c13_bssfp_pulse_meets_its_magnetisation_spec_on_the_oracle_chain():
    """bSSFP_pulse_sb_mb.m (lactate selected, 58 taps): spec -> fir_ap_cvx -> reverse -> b2rf -> abrm -> |Mxy| inside
    rf_spec's bands.  The allowance over the ripple is the hard-pulse model error of the 58-sample pulse."""
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    n, T = 100, 4.0
    dt = T / n
    cf = mbfir.spec.spectrum_c13(14.0)[[5, 0, 2, 3, 1]] * 1e-3
    cf = cf - cf[4]
    FA, rp = [0, 0, 0, 0, 60], [.005] * 4 + [.01]
    f, a, d = mbfir.spec.band_spec(n, dt, list(cf), [0.1] * 5, FA, rp, "ex")
    h, s = designers.fir_ap_cvx(58, f, a, d, 0.1, 1e-3)
    assert s == "Solved"
    rf = slr.b2rf(h[::-1])                                   # dzrf_mb.m:220,239-240
    fs = 1 / dt
    fk = np.linspace(-fs / 2, fs / 2, 2001)[:-1]
    aa, bb = bloch.abrm(rf, fk * len(rf) * dt)
    mxy = np.abs(bloch.mxy_excitation(aa, bb))
    for i in range(5):
        sel = (fk >= f[2 * i] * fs / 2) & (fk <= f[2 * i + 1] * fs / 2)
        assert sel.sum() >= 5
        assert np.max(np.abs(mxy[sel] - np.sin(np.radians(FA[i])))) <= rp[i] * 1.1


def test_fir_upsample_keeps_the_spectrum_in_band():
    """fir_upsample.m: resample by p then / p -- the upsampled taps have the same spectrum over the original band
    and (nearly) nothing in the images."""
    rng = np.random.default_rng(2)
    h = (rng.standard_normal(40) + 1j * rng.standard_normal(40)) * np.hanning(40)
    # band-limit the test signal to 60 % of Nyquist so that the images are clean
    H = np.fft.fft(h, 256); H[int(0.3 * 256):256 - int(0.3 * 256)] = 0; h = np.fft.ifft(H)[:40] * np.hanning(40)
    u = mbfir.fir_upsample(h, 0.04, 0.02)
    assert len(u) == 80
    w = np.linspace(-0.25, 0.25, 41) * np.pi                # inside the original band, on the new axis
    Hh = (h[None, :] * np.exp(-2j * w[:, None] * np.arange(40)[None, :])).sum(1)
    Hu = (u[None, :] * np.exp(-1j * w[:, None] * np.arange(80)[None, :])).sum(1)
    assert np.max(np.abs(np.abs(Hu) - np.abs(Hh))) < 0.02 * np.max(np.abs(Hh))
    assert np.array_equal(mbfir.fir_upsample(h, 0.04, 0.04), h)


def test_rf_mrange_desired():
    assert np.allclose(mbfir.rf_mrange_desired(60, 0.01, "ex"), (np.sin(np.pi / 3) - 0.01, np.sin(np.pi / 3) + 0.01))
    assert np.allclose(mbfir.rf_mrange_desired(90, 0.01, "ex"), (0.99, 1.0))
    assert np.allclose(mbfir.rf_mrange_desired(0, 0.005, "sat"), (0.995, 1.0))
    assert np.allclose(mbfir.rf_mrange_desired(180, 0.02, "inv"), (-1.0, -0.98))
    assert np.allclose(mbfir.rf_mrange_desired(180, 0.02, "se"), (0.98, 1.0))
    with pytest.raises(ValueError):
        mbfir.rf_mrange_desired(200, 0.01, "ex")
    with pytest.raises(ValueError):
        mbfir.rf_mrange_desired(20, 0.01, "xx")


def test_dzrf_mb_argument_errors_need_no_gpu():
    """dzrf_mb.m:56-98,136-157: the checks that run before any design."""
    cf = [[-1.0], [1.5]]
    with pytest.raises(ValueError, match="nucleus"):
        mbfir.dzrf_mb(128, 0.05, cf, [0.4, 0.4], [0, 90], [0.01, 0.01], "sat", "ap_cvx", "N-15")
    with pytest.raises(ValueError, match="not an integer"):
        mbfir.dzrf_mb(127, 0.05, cf, [0.4, 0.4], [0, 90], [0.01, 0.01], "sat", "ap_cvx", "H-1", 0, 2)
    with pytest.raises(ValueError, match="shift_f"):
        mbfir.dzrf_mb(128, 0.05, cf, [0.4, 0.4], [0, 90], [0.01, 0.01], "sat", "ap_cvx", "H-1", 0, 1, None, 0, None, None, 3)
    with pytest.raises(NameError, match="TBW"):
        mbfir.dzrf_mb(128, 0.05, cf, [0.4, 0.4], [0, 90], [0.01, 0.01], "sat", "ms", "H-1")
    with pytest.raises(ValueError, match="sampling rate"):
        mbfir.dzrf_mb(128, 0.5, cf, [0.4, 0.4], [0, 90], [0.01, 0.01], "sat", "ap_cvx", "H-1")      # fs/2 = 1 kHz < 1.7 kHz
    with pytest.raises(ValueError, match="broken in the reference"):
        mbfir.dzrf_mb(128, 0.05, cf, [0.4, 0.4], [0, 90], [0.01, 0.01], "st", "ap_cvx", "H-1")
