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


# ---- blochsimfz restatement (oracle/bloch.py; bloch_simulation/blochC.c:171-236,283-512) -----------------
def test_calcrotmat_is_a_proper_rotation_about_its_axis():
    rng = np.random.default_rng(1)
    n = rng.standard_normal((50, 3)) * rng.uniform(0, 4, (50, 1))
    R = bloch.calcrotmat(n[:, 0], n[:, 1], n[:, 2])
    assert np.abs(R @ np.transpose(R, (0, 2, 1)) - np.eye(3)).max() <= 1e-14
    assert np.abs(np.linalg.det(R) - 1).max() <= 1e-14
    assert np.abs(np.einsum("kij,kj->ki", R, n) - n).max() <= 1e-13          # the axis is fixed
    assert np.array_equal(bloch.calcrotmat(0.0, 0.0, 0.0), np.eye(3))         # blochC.c:182-193
    # right-handed rotation by |n| about n (rmat[1] = -arai2 = +sin for n = z); off-resonance enters as rotz = -2 pi df dt
    Rz = bloch.calcrotmat(0.0, 0.0, 0.3)
    assert np.allclose(Rz @ [1, 0, 0], [np.cos(0.3), np.sin(0.3), 0])


def test_blochsimfz_free_precession_and_relaxation_are_analytic():
    nt, ts, t1, t2 = 80, 0.5e-3, 0.3, 0.05
    df = np.array([0.0, 10.0, -35.0, 400.0])
    m0 = np.zeros((4, 1, 3))
    m0[:, 0, 0] = 1.0
    m = bloch.blochsimfz(np.zeros(nt), None, ts, t1, t2, df, np.zeros((1, 3)), 0, m0)
    T = nt * ts
    ph = -df * bloch.TWOPI_REF * T
    assert np.abs(m[:, 0, 0, 0] - np.exp(-T / t2) * np.cos(ph)).max() <= 1e-13
    assert np.abs(m[:, 0, 0, 1] - np.exp(-T / t2) * np.sin(ph)).max() <= 1e-13
    assert np.abs(m[:, 0, 0, 2] - (1 - np.exp(-T / t1))).max() <= 1e-13
    # a gradient acts like an off-resonance gamma G x / TWOPI
    G, x = 0.2, 1.7
    mg = bloch.blochsimfz(np.zeros(nt), np.full((nt, 1), G), ts, t1, t2, [0.0], np.array([[x, 0, 0]]), 0, m0[:1])
    mf = bloch.blochsimfz(np.zeros(nt), None, ts, t1, t2, [bloch.GAMMA_C13 * G * x / bloch.TWOPI_REF], np.zeros((1, 3)), 0, m0[:1])
    assert np.abs(mg - mf).max() <= 1e-12


def test_blochsimfz_without_relaxation_is_the_cayley_klein_simulation():
    """T1 = T2 -> infinity: the same rotation per sample as rf_tools/abrm.m, so Mxy = 2 a conj(b), Mz = 1 - 2 |b|^2
    (abr.m:11-12 up to the conjugation of the two simulators' axis conventions)."""
    rng = np.random.default_rng(0)
    n = 64
    rf = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.05          # radians per sample
    dt = 4e-3 / n
    dfs = np.linspace(-800, 800, 11)
    m = bloch.blochsimfz(rf / (bloch.GAMMA_C13 * dt), None, dt, 1e12, 1e12, dfs, np.zeros((1, 3)), 0)
    a, b = bloch.abrm(rf, np.ones(n) * dt * bloch.TWOPI_REF, dfs)
    assert np.abs(m[:, 0, 0, 0] + 1j * m[:, 0, 0, 1] - 2 * a * np.conj(b)).max() <= 1e-11
    assert np.abs(m[:, 0, 0, 2] - (1 - 2 * np.abs(b) ** 2)).max() <= 1e-11


def test_blochsimfz_modes():
    """mode 2 records every sample and ends at the mode-0 endpoint; mode 1 returns the periodic steady state
    (one more period reproduces it); mode 3 = that steady state followed through the period."""
    rng = np.random.default_rng(3)
    nt = 40
    b1 = (rng.standard_normal(nt) + 1j * rng.standard_normal(nt)) * 0.02
    gr = rng.standard_normal((nt, 3)) * 0.1
    ts = rng.uniform(0.5e-4, 2e-4, nt)
    df = np.array([-120.0, 0.0, 77.0])
    pos = rng.standard_normal((2, 3))
    args = (b1, gr, ts, 0.08, 0.03, df, pos)
    m0 = bloch.blochsimfz(*args, 0)
    m2 = bloch.blochsimfz(*args, 2)
    assert m0.shape == (3, 2, 1, 3) and m2.shape == (3, 2, nt, 3)
    assert np.abs(m2[:, :, -1] - m0[:, :, 0]).max() <= 1e-15
    ss = bloch.blochsimfz(*args, 1)
    again = bloch.blochsimfz(*args, 0, ss[:, :, 0])
    assert np.abs(again - ss).max() <= 1e-13
    m3 = bloch.blochsimfz(*args, 3)
    assert np.abs(m3[:, :, -1] - ss[:, :, 0]).max() <= 1e-13
    assert np.abs(m3 - bloch.blochsimfz(*args, 2, ss[:, :, 0])).max() <= 1e-15
