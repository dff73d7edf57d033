"""Device forward simulation (mbfir_abr) and the dzrf_mb driver end to end on the GPU."""
import json
import os

import numpy as np
import pytest

import mbfir
from oracle import bloch, slr

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "slr_golden.json")))


def cx(d):
    return np.array(d["re"]) + 1j * np.array(d["im"])


@pytest.mark.parametrize("name", ["ap_c13_64_p071", "sinc127_p026", "rand511_p071", "one_tap_p097", "lin_cplx31_p097"])
def test_device_abrm_against_the_oracle(name):
    b = cx(GOLD[name]["b"])
    rf = slr.b2rf(b)
    x = np.linspace(-len(b) / 2, len(b) / 2, 777)
    a0, b0 = bloch.abrm(rf, x)
    a1, b1 = mbfir.abrm(rf, x)
    assert np.max(np.abs(a1 - a0)) <= 1e-12 and np.max(np.abs(b1 - b0)) <= 1e-12
    g = np.linspace(0.5, 1.5, len(rf)) * 2 * np.pi / len(rf)          # explicit per-sample weights
    a0, b0 = bloch.abrm(rf, g, x)
    a1, b1 = mbfir.abrm(rf, g, x)
    assert np.max(np.abs(a1 - a0)) <= 1e-12 and np.max(np.abs(b1 - b0)) <= 1e-12
    a2, b2 = mbfir.abr(rf, x)
    ao, bo = bloch.abr(rf, x)
    assert np.max(np.abs(b2 - bo)) <= 1e-12 and np.max(np.abs(a2 - ao)) <= 1e-12


@pytest.mark.parametrize("name", ["ap_c13_64_p071", "sinc127_p026", "rand200_p071", "qp_modelA48_p026"])
def test_device_closed_loop_beta_to_rf_to_beta(name):
    """b -> mbfir.b2rf -> mbfir.abrm(hard_pulse) : |beta(x)| = |B(-2 pi x / n)| to rounding, all on the device."""
    b = cx(GOLD[name]["b"])
    n = len(b)
    rf = mbfir.b2rf(b)
    x = np.linspace(-n / 2, n / 2, 513)[:-1]
    a1, b1 = mbfir.abrm(rf, x, hard_pulse=True)
    ah, bh = bloch.hard_pulse_ab(rf, x)
    assert np.max(np.abs(a1 - ah)) <= 1e-12 and np.max(np.abs(b1 - bh)) <= 1e-12
    w = -2 * np.pi * x / n
    B = (b[None, :] * np.exp(1j * w[:, None] * np.arange(n)[None, :])).sum(1)
    assert np.max(np.abs(np.abs(b1) - np.abs(B))) < 1e-10
    assert np.max(np.abs(np.abs(a1) ** 2 + np.abs(b1) ** 2 - 1)) < 1e-12


def c13_args():
    cf = mbfir.spec.spectrum_c13(14.0)[[5, 0, 2, 3, 1]] * 1e-3
    cf = cf - cf[4]
    return list(cf), [0.1] * 5, [0, 0, 0, 0, 60], [.005] * 4 + [.01]


def check_profile(rf_pulse, dt, gamma, rf_spec, slack=1.1, hard_pulse=True):
    """|Mxy| over the bands of rf_spec.  hard_pulse: the model the SLR design is exact in -- the band limits then hold
    at the designer's grid points (fir_ap_cvx.m:44-52: 15 n samples, 3-4 per 100 Hz band here) and overshoot by a
    few per cent of the ripple between them, hence the slack; otherwise abrm.m's joint rotations, the reference's simulator, which differs from it
    by O(rf * om) per sample -- a few 1e-3 at these pulse lengths."""
    rf = rf_pulse * (2 * np.pi * gamma * dt)                              # Gauss -> radians per sample (rfscaleg.m)
    fs = 1 / dt
    fk = np.linspace(-fs / 2, fs / 2, 4001)[:-1]
    x = fk * len(rf) * dt
    a, b = mbfir.abrm(rf, -x, hard_pulse=True) if hard_pulse else mbfir.abrm(rf, x)
    mxy = np.abs(2 * np.conj(a) * b)
    f = np.asarray(rf_spec["f"]) * fs / 2
    for i in range(len(rf_spec["d"])):
        sel = (fk >= f[2 * i]) & (fk <= f[2 * i + 1])
        assert sel.sum() >= 5
        assert np.max(np.abs(mxy[sel] - rf_spec["a"][2 * i])) <= rf_spec["d"][i] * slack + 1e-6
    return mxy


def test_dzrf_mb_c13_bssfp_pulse_end_to_end():
    """bSSFP_pulse_sb_mb.m:54: dzrf_mb(100, 0.04, ..., 'ex', 'ap_minorder_cvx', 'C-13', 0, 1, [], dbg, 58) on the
    device, then the simulated |Mxy| against the returned rf_spec (what the script plots)."""
    cf, rng, FA, rp = c13_args()
    rf_pulse, b, rf_spec, b_spec = mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", "ap_minorder_cvx", "C-13", 0, 1, None, 0, 58)
    assert len(rf_pulse) == len(b) == 58
    assert np.allclose(rf_spec["a"], [0] * 8 + [np.sin(np.pi / 3)] * 2) and np.allclose(rf_spec["d"], rp)
    check_profile(rf_pulse, 0.04, 1.0705, rf_spec)
    check_profile(rf_pulse, 0.04, 1.0705, rf_spec, slack=1.15, hard_pulse=False)       # the reference's simulator
    # the same pulse through the oracle chain
    from oracle import designers
    f, a, d = b_spec["f"], b_spec["a"], b_spec["d"]
    ho, so = designers.fir_ap_cvx(58, f, a, d, 0.1, 1e-3)
    ref = slr.rfscaleg(slr.b2rf(ho[::-1]), 58 * 0.04, 1.0705)
    assert np.max(np.abs(rf_pulse - ref)) <= 1e-6 * np.max(np.abs(ref))
    assert 0.05 < np.max(np.abs(rf_pulse)) < 1.0                          # Gauss, a sane 60-degree C-13 pulse


def test_dzrf_mb_filter_types_and_options():
    cf, rng, FA, rp = c13_args()
    # plain ap_cvx at full length, minimum-order search (min_order = 0.9 default) and fixed transition widening
    for ftype, kw in (("ap_cvx", {}), ("ap_minstopripple_cvx", {}), ("ap_minorder_cvx", dict(probes=4)), ("ap_mintran_cvx", dict(probes=3))):
        rf_pulse, b, rf_spec, b_spec = mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", ftype, "C-13", **kw)
        assert len(rf_pulse) == len(b) and 50 <= len(b) <= 100, ftype
        # 1.4: the 62-tap design's S(w) dips to -8e-5 between grid points, fmp2 takes abs() (fir_ap_cvx.m:281) and
        # |H|^2 then misses S by 2.6e-3 -- in the oracle exactly as on the device (tools/gpu_dzrf.py)
        check_profile(rf_pulse, 0.04, 1.0705, rf_spec, slack=1.4)
        if ftype == "ap_minorder_cvx":
            assert len(b) == int(np.ceil(100 * 0.1 + 57 * 0.9))              # fir_ap.m:173, min order 57
        if ftype == "ap_mintran_cvx":
            assert np.all(np.diff(b_spec["f"])[0::2] > 0.1 / 12.5)            # bands widened
    # saturation pulse type and the frequency shift option (shift the 90-degree band to f = 0, design, shift back)
    rf_pulse, b, rf_spec, b_spec = mbfir.dzrf_mb(128, 0.05, [[-1.0], [1.5]], [0.4, 0.4], [0, 90], [0.01, 0.01], "sat", "ap_cvx", "H-1", 0, 1,
                                                 None, 0, None, None, 2)
    rf = rf_pulse * (2 * np.pi * 4.2576 * 0.05)
    fk = np.array([-1.1, -1.0, -0.9, 1.4, 1.5, 1.6])
    a, bb = mbfir.abrm(rf, fk * len(rf) * 0.05)
    mz = 1 - 2 * np.abs(bb) ** 2
    assert np.all(mz[:3] > 0.985) and np.all(np.abs(mz[3:]) < 0.035)       # abrm.m's model, O(rf om) off the SLR one
    a, bb = mbfir.abrm(rf, -fk * len(rf) * 0.05, hard_pulse=True)
    mz = 1 - 2 * np.abs(bb) ** 2
    assert np.all(mz[:3] > 0.99 - 1e-3) and np.all(np.abs(mz[3:]) < 0.0125)
    # errors and the failure return
    with pytest.raises(NameError):
        mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", "ms")
    with pytest.raises(ValueError):
        mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", "ap_cvx", "N-15")
    with pytest.raises(ValueError):
        mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", "ap_cvx", "C-13", 0, 3)
    # flip_zero = 1: pass-band zeros reflected for a lower peak of beta, same |B(w)| -- the profile still meets the spec
    rf0, b0, _, _ = mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", "ap_minorder_cvx", "C-13", 0, 1, None, 0, 58)
    rf1, b1, rf_spec, _ = mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", "ap_minorder_cvx", "C-13", 1, 1, None, 0, 58)
    assert len(b1) == 58 and np.max(np.abs(b1)) <= np.max(np.abs(b0)) * (1 + 1e-9)
    check_profile(rf1, 0.04, 1.0705, rf_spec, slack=1.15)
    rf_pulse, b, rf_spec, b_spec = mbfir.dzrf_mb(100, 0.04, cf, rng, FA, rp, "ex", "ap_minorder_cvx", "C-13", 0, 1, None, 0, 40)
    assert len(rf_pulse) == 0 and len(b) == 0 and len(rf_spec["d"]) == 5      # 40 taps: 'Filter design failed.'


def test_dzrf_mb_downsampled_design():
    """downsampling = 2: design at 80 taps / 0.08 ms (the spec still fits the halved bandwidth of 6.25 kHz),
    upsample to 160 / 0.04 ms (dzrf_mb.m:92-98,228-231)."""
    cf, rng, FA, rp = c13_args()
    rf_pulse, b, rf_spec, b_spec = mbfir.dzrf_mb(160, 0.04, cf, rng, FA, rp, "ex", "ap_cvx", "C-13", 0, 2)
    assert len(rf_pulse) == len(b) == 160
    check_profile(rf_pulse, 0.04, 1.0705, rf_spec, slack=1.5)                         # resample's pass-band droop


# ---- mbfir_bloch: the device twin of bloch_simulation/blochC.c (blochsimfz) --------------------------------
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_device_bloch_matches_the_oracle_in_every_mode(mode):
    rng = np.random.default_rng(10 + mode)
    nt = 300                                                 # more than one LDS chunk of 256 samples
    b1 = (rng.standard_normal(nt) + 1j * rng.standard_normal(nt)) * 0.02
    gr = rng.standard_normal((nt, 3)) * 0.1
    ts = rng.uniform(0.5e-4, 2e-4, nt)
    df = np.linspace(-500, 500, 37)
    pos = rng.standard_normal((5, 3))
    m0 = rng.standard_normal((37, 5, 3)) * 0.3
    ref = bloch.blochsimfz(b1, gr, ts, 0.08, 0.03, df, pos, mode, m0)
    mx, my, mz = mbfir.bloch(b1, gr, ts, 0.08, 0.03, df, pos, mode, m0[..., 0], m0[..., 1], m0[..., 2])
    got = np.stack([mx, my, mz], -1).reshape(ref.shape)
    assert np.abs(got - ref).max() <= 1e-12
    # end times instead of intervals (blochC.c:669-681) and the H-1 constant (blochH.c:6)
    refh = bloch.blochsimfz(b1, gr[:, :1], ts, 0.5, 0.1, df, pos[:, :1], mode & 2, gamma=bloch.GAMMA_H1)
    mx, my, mz = mbfir.bloch(b1, gr[:, 0], np.cumsum(ts), 0.5, 0.1, df, pos[:, 0], mode & 2, nucleus="H-1")
    assert np.abs(np.stack([mx, my, mz], -1).reshape(refh.shape) - refh).max() <= 1e-12


def test_device_bloch_closes_the_loop_with_abr_as_sim_rf_spectral_runs_it():
    """sim_rf_spectral.m:63-78 simulates the designed pulse with T1 = T2 = 1e3 s over 2048 off-resonances.  Without
    relaxation blochsimfz applies the same rotation per sample as abrm.m, so |Mxy| = 2 |a b| and Mz = 1 - 2 |b|^2 of
    mbfir_abr; with T1 = T2 = 1e3 s and a 4 ms pulse the relaxation moves that by 4e-6."""
    n = 58
    f, a, d = mbfir.spec.spec_c13_bssfp(n) if hasattr(mbfir.spec, "spec_c13_bssfp") else (None, None, None)
    h, status = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3)
    assert status == "Solved"
    rf = np.asarray(mbfir.b2rf(h[::-1])).ravel()                  # radians per sample (dzrf_mb.m:220,239-240)
    dt = 4e-3 / n
    b1 = rf / (mbfir.GAMMA_C13 * dt)                              # Gauss (rfscaleg)
    df = np.linspace(-3000, 3000, 2048)                           # N_simu = 256*8 (sim_rf_spectral.m:63)
    av, bv = mbfir.abrm(rf, np.ones(n) * dt * 6.283185, df)
    for T, tol in ((1e3, 1e-5), (1e12, 1e-10)):
        mx, my, mz = mbfir.bloch(b1, np.zeros(n), dt, T, T, df, 0.0, 0)
        assert mx.shape == (2048, 1)
        assert np.abs(mx[:, 0] + 1j * my[:, 0] - 2 * av * np.conj(bv)).max() <= tol
        assert np.abs(mz[:, 0] - (1 - 2 * np.abs(bv) ** 2)).max() <= tol
