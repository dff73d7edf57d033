"""Spec front-end: physical multiband description -> normalised (f, a, d) band spec.

Host-side mirror of the reference's spec helpers, plain Python, no GPU:
  spectrum_c13   <- spectrum_C13.m:27-40
  dinf           <- dinf.m:9-29 (Parks-McClellan D-infinity fit)
  rf_bandedge    <- rf_bandedge.m:27-161
  rf_ripple_gfa  <- rf_ripple_GFA.m:180-297 (the exact 'asin' mapping; appro=0 is what dzrf_mb.m:106 uses)
  band_spec      <- dzrf_mb.m:103-160 (band edges, beta-polynomial amplitudes/ripples, frequency shift)
The two specs the reference's driver scripts define, used by the tests and the benchmark:
  spec_c13_bssfp   <- bSSFP_pulse_sb_mb.m:9-52   (S-C13 of SURVEY 8c)
  spec_h1_dualband <- specsat_H1_dualband.m:5-47 (S-H1)
"""
import math

import numpy as np


def spectrum_c13(B0):
    """Resonance offsets (Hz, relative to pyruvate) of pyruvate, lactate, alanine, pyruvate hydrate,
    bicarbonate, urea at field B0 (T)."""
    gamma = 10.705e6
    cs = np.array([170.60, 182.98, 176.32, 178.91, 160.9, 163.13])
    f0 = gamma * B0 * (1 + cs * 1e-6)
    return f0 - f0[0]


def dinf(d1, d2):
    a1, a2, a3, a4, a5, a6 = 5.309e-3, 7.114e-2, -4.761e-1, -2.66e-3, -5.941e-1, -4.278e-1
    l1, l2 = math.log10(d1), math.log10(d2)
    return (a1 * l1 * l1 + a2 * l1 + a3) * l2 + (a4 * l1 * l1 + a5 * l1 + a6)


def _fa2beta(rfa_r, rfa_l, FA):
    if rfa_r > math.pi:                                     # rf_ripple_GFA.m:281-286
        lo = min(math.sin(rfa_l / 2), math.sin(rfa_r / 2))
        return lo, 1.0
    return math.sin(rfa_l / 2), math.sin(rfa_r / 2)


def rf_ripple_gfa(FA_deg, ripple_M, ptype):
    """Range [min, max] of |B_N| that keeps the magnetisation within +-ripple_M of its nominal value
    for a band of flip angle FA_deg (rf_ripple_GFA.m, rf_ripple_asin)."""
    if not 0 <= FA_deg <= 180:
        raise ValueError("Flip angle should be in the range of [0 180] degree")
    FA = FA_deg * math.pi / 180
    if ptype == "ex":
        if math.sin(FA) + ripple_M >= 1:
            rfa = math.asin(math.sin(FA) - ripple_M)
            rfa_l, rfa_r = rfa, math.pi - rfa
        elif FA <= math.pi / 2:
            rfa_l, rfa_r = math.asin(math.sin(FA) - ripple_M), math.asin(math.sin(FA) + ripple_M)
        else:
            rfa_r, rfa_l = math.pi - math.asin(math.sin(FA) - ripple_M), math.pi - math.asin(math.sin(FA) + ripple_M)
        return _fa2beta(rfa_r, rfa_l, FA)
    if ptype in ("sat", "inv"):
        if math.cos(FA) + ripple_M > 1:
            rfa = math.acos(math.cos(FA) - ripple_M)
            rfa_l, rfa_r = -rfa, rfa
        elif math.cos(FA) - ripple_M < -1:
            rfa = math.acos(math.cos(FA) + ripple_M)
            rfa_l, rfa_r = rfa, 2 * math.pi - rfa
        else:
            rfa_r, rfa_l = math.acos(math.cos(FA) - ripple_M), math.acos(math.cos(FA) + ripple_M)
        return _fa2beta(rfa_r, rfa_l, FA)
    if ptype == "se":
        mid = math.sin(FA / 2) ** 2
        return math.sqrt(max(0.0, min(1.0, mid - ripple_M))), math.sqrt(max(0.0, min(1.0, mid + ripple_M)))
    if ptype == "st":
        raise ValueError("ptype 'st' is broken in the reference (undefined mid_M, rf_ripple_GFA.m:201-208)")
    raise ValueError("Unrecognized Pulse Type -- %s" % ptype)


def rf_bandedge(n, dt, mb_cf, mb_range, mb_FA, mb_ripple, ptype):
    """Normalised band edges in [-1, 1].  mb_cf: per band a centre frequency or a (lo, hi) range in kHz;
    mb_range: per-band width in kHz, or None to derive the edges from the D-infinity transition width."""
    T, fs, k = n * dt, 1.0 / dt, len(mb_FA)
    lohi = [(float(np.ravel(c)[0]), float(np.ravel(c)[-1])) for c in mb_cf]
    f = np.zeros(2 * k)
    if mb_range is None:
        FA = np.asarray(mb_FA, dtype=float)
        hi = {"st": 90, "ex": 90, "sat": 90, "inv": 180, "se": 180}[ptype]
        d1 = mb_ripple[int(np.argmin(np.abs(FA - hi)))]
        d2 = mb_ripple[int(np.argmin(np.abs(FA)))]
        delta1, delta2 = {"st": (math.sqrt(d1 / 2), d2 / math.sqrt(2)), "ex": (math.sqrt(d1 / 2), d2 / math.sqrt(2)),
                          "inv": (d1 / 8, math.sqrt(d2 / 2)), "sat": (d1 / 2, math.sqrt(d2)),
                          "se": (d1 / 4, math.sqrt(d2))}[ptype]
        df = dinf(delta1, delta2) / T
        for i, (lo, hi_) in enumerate(lohi):
            f[2 * i], f[2 * i + 1] = lo, hi_
        f1 = f.copy()
        for i in range(k - 1):
            f1[2 * i + 1] = f1[2 * i + 2] = (f[2 * i + 1] + f[2 * i + 2]) / 2
        f1[0] = f[0] - (f1[1] - f[1])
        f1[-1] = f[-1] + (f[-2] - f1[-2])
        f = f1.copy()
        f[0::2] += df / 2
        f[1::2] -= df / 2
    else:
        for i, (lo, hi_) in enumerate(lohi):
            f[2 * i], f[2 * i + 1] = lo - mb_range[i] / 2, hi_ + mb_range[i] / 2
    if np.any(np.diff(f) < 0):
        raise ValueError("Incompatible spec of frequency range: f is not monotonically increasing")
    if f[0] < -fs / 2 or f[-1] > fs / 2:
        raise ValueError("the sampling rate is not enough, increase n")
    return f / (fs / 2)


def band_spec(n, dt, mb_cf, mb_range, mb_FA, mb_ripple, ptype, shift_f=0):
    """(f, a, d) handed to the FIR designers by dzrf_mb.m:103-160."""
    f = rf_bandedge(n, dt, mb_cf, mb_range, mb_FA, mb_ripple, ptype)
    a, d = np.zeros(2 * len(mb_FA)), np.zeros(len(mb_FA))
    for i, (fa, rp) in enumerate(zip(mb_FA, mb_ripple)):
        lo, hi = rf_ripple_gfa(fa, rp, ptype)
        a[2 * i] = a[2 * i + 1] = (hi + lo) / 2
        d[i] = (hi - lo) / 2
    if shift_f == 1:                                        # centre of the high-flip-angle bands to f = 0
        idx = [i for i, fa in enumerate(mb_FA) if fa > 60]
        f = f - (f[2 * idx[0]] + f[2 * idx[-1] + 1]) / 2
    elif shift_f == 2:                                      # highest flip angle band to f = 0
        i = int(np.argmax(mb_FA))
        f = f - (f[2 * i] + f[2 * i + 1]) / 2
    elif shift_f != 0:
        raise ValueError("shift_f = %r is not an option. Options are 0,1,2" % (shift_f,))
    return f, a, d


def spec_c13_bssfp(n=100, T=4.0, B0=14.0, FA=60.0, d1=0.01, d2=0.005):
    """bSSFP C-13 multiband excitation, lactate selected (bSSFP_pulse_sb_mb.m:9-52): bands urea, pyruvate,
    alanine, pyruvate hydrate, lactate; 0.1 kHz wide; duration T ms over n samples."""
    cf = spectrum_c13(B0)[[5, 0, 2, 3, 1]] * 1e-3           # kHz
    cf = cf - cf[4]
    return band_spec(n, T / n, list(cf), [0.1] * 5, [0, 0, 0, 0, FA], [d2, d2, d2, d2, d1], "ex")


def spec_h1_dualband(n=260, T=26.0):
    """Dual-band H-1 spectral saturation at 3 T (specsat_H1_dualband.m:5-47): bands 1.8-2.5, 3-4.1 and
    4.8-5.4 ppm, flip angles 120 / 0 / 90 degrees, ripples 0.05 / 0.001 / 0.05, dt a multiple of 4 us."""
    B0 = 127794577 / (42.577 * 1e6)
    dt = T / n
    if (dt / 4e-3) % 1 != 0:
        dt = 4e-3 * math.floor(dt / 4e-3)
    bands = [(1.8, 2.5), (3.0, 4.1), (4.8, 5.4)]
    ref = sum(bands[2]) / 2
    cf = [((lo - ref) * B0 * 42.577e-3, (hi - ref) * B0 * 42.577e-3) for lo, hi in bands]
    return band_spec(n, dt, cf, [0.01] * 3, [120, 0, 90], [0.05, 0.001, 0.05], "sat", shift_f=1)


def spec_rand(n, seed, kmin=2, kmax=8):
    """S-RAND (no reference script -- the robustness / heterogeneous-batch workload): k in {kmin..kmax}
    non-overlapping bands on [-1, 1], widths U(0.01, 0.1), gaps >= 8 / n, |B_N| amplitudes 0 or U(0.2, 0.9) (at least one
    pass band), ripples U(0.002, 0.02); NumPy default_rng(12345 + seed).  Returns (f, a, d) as the FIR designers take them;
    feasibility at a given order is for the caller to establish (the designers return 'Failed' on an infeasible draw)."""
    rng = np.random.default_rng(12345 + int(seed))
    k = int(rng.integers(kmin, kmax + 1))
    widths = rng.uniform(0.01, 0.1, k)
    gmin = 8.0 / n
    slack = 2.0 - widths.sum() - (k + 1) * gmin
    if slack <= 0:
        raise ValueError("spec_rand: %d bands with gaps of 8 / n do not fit at n = %d" % (k, n))
    cuts = np.sort(rng.uniform(0.0, slack, k + 1))
    extra = np.diff(np.concatenate([[0.0], cuts]))           # the free room dealt over the k + 1 gaps (the rest stays at the right end)
    f, x = np.zeros(2 * k), -1.0
    for i in range(k):
        x += gmin + extra[i]
        f[2 * i], f[2 * i + 1] = x, x + widths[i]
        x += widths[i]
    amp = np.where(rng.random(k) < 0.5, 0.0, rng.uniform(0.2, 0.9, k))
    if not np.any(amp > 0):
        amp[int(rng.integers(0, k))] = 0.6
    return f, np.repeat(amp, 2), rng.uniform(0.002, 0.02, k)
