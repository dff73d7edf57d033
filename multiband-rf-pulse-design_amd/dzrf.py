"""dzrf_mb.m, host side: physical multiband spec -> beta polynomial (device designers) -> inverse SLR (device)
-> RF pulse in Gauss.  Mirrors the reference's signature, defaults, option strings and quirks; plots are dropped.

What is not carried over, and says so when asked for: ftype 'ms' (undefined TBW in the reference, dzrf_mb.m:165).
"""
import math

import numpy as np

from . import spec as _spec

GAMMA = {"H-1": 4.2576, "C-13": 1.0705}          # kHz/G (dzrf_mb.m:81-88)


def rf_mrange_desired(FA_deg, ripple_M, ptype):
    """rf_Mrange_desired.m:20-47: range of the magnetisation component the pulse type controls."""
    if not 0 <= FA_deg <= 180:
        raise ValueError("Flip angle should be in the range of [0 180] degree")
    FA = FA_deg * math.pi / 180
    if ptype in ("st", "ex"):
        return math.sin(FA) - ripple_M, min(math.sin(FA) + ripple_M, 1.0)
    if ptype in ("sat", "inv"):
        return max(math.cos(FA) - ripple_M, -1.0), min(math.cos(FA) + ripple_M, 1.0)
    if ptype == "se":
        m = math.sin(FA / 2) ** 2
        return max(m - ripple_M, 0.0), min(m + ripple_M, 1.0)
    raise ValueError("Unrecognized Pulse Type -- %s" % ptype)


def fir_upsample(h, dt1, dt2):
    """fir_upsample.m:19-21: `resample(h, n, 1, floor(length(h)/2)) / n`, n = round(dt1/dt2).  MATLAB's resample
    is restated from its documentation: anti-aliasing FIR of length 2 N p + 1 (N = the 4th argument), the
    least-squares ideal low-pass with cut-off 1/(2p) (a truncated sinc) under a Kaiser window with beta = 5,
    gain p, applied to the zero-stuffed input with its group delay removed."""
    h = np.asarray(h, dtype=np.complex128).ravel()
    p = int(round(dt1 / dt2))
    if p < 2:
        return h.copy()
    N = len(h) // 2
    L = 2 * N * p + 1
    k = np.arange(L) - (L - 1) / 2
    taps = np.sinc(k / p) / p * np.kaiser(L, 5.0)
    taps = p * taps / taps.sum()
    up = np.zeros(len(h) * p, dtype=np.complex128)
    up[::p] = h
    y = np.convolve(up, taps)
    d = (L - 1) // 2
    return y[d:d + len(h) * p] / p


def dzrf_mb(n, dt, mb_cf, mb_range, mb_FA, mb_ripple, ptype="sat", ftype="ap_cvx", nucleus="C-13", flip_zero=0,
            downsampling=1, Peak=1e-3, dbg=0, min_order=0.9, min_tran=0.85, shift_f=0, name_cell=None, *,
            opts=None, probes=1):
    """Returns (rf_pulse, b, rf_spec, b_spec) like dzrf_mb.m; ([], [], rf_spec, b_spec) when the filter design
    fails (dzrf_mb.m:216-218).  rf_pulse in Gauss, dt in ms, frequencies in kHz."""
    import mbfir
    nucleus = nucleus or "C-13"
    if nucleus not in GAMMA:
        raise ValueError("No such option for nucleus. Options are H-1 and C-13")
    gamma = GAMMA[nucleus]
    downsampling = downsampling or 1                                    # isempty -> default (dzrf_mb.m:62-71)
    flip_zero = flip_zero or 0
    Peak = 1e-3 if Peak is None else Peak
    min_order = 0.9 if min_order is None else min_order
    min_tran = 0.85 if min_tran is None else min_tran
    shift_f = shift_f or 0
    nd = n / downsampling                                               # dzrf_mb.m:92-98
    if abs(nd - round(nd)) > 1e-10:
        raise ValueError("n/downsampling is not an integer")
    n, dt = int(round(nd)), dt * downsampling
    fs = 1.0 / dt
    mb_FA, mb_ripple = list(mb_FA), list(mb_ripple)
    f = _spec.rf_bandedge(n, dt, mb_cf, mb_range, mb_FA, mb_ripple, ptype)                  # :101
    a, d = np.zeros(2 * len(mb_FA)), np.zeros(len(mb_FA))
    a_M, d_M = np.zeros(2 * len(mb_FA)), np.zeros(len(mb_FA))
    for i, (fa, rp) in enumerate(zip(mb_FA, mb_ripple)):                                     # :104-121
        lo, hi = _spec.rf_ripple_gfa(fa, rp, ptype)
        a[2 * i] = a[2 * i + 1] = (hi + lo) / 2
        d[i] = (hi - lo) / 2
        lo, hi = rf_mrange_desired(fa, rp, ptype)
        a_M[2 * i] = a_M[2 * i + 1] = (hi + lo) / 2
        d_M[i] = (hi - lo) / 2
    b_spec = dict(f=f / downsampling, a=a, d=d)
    rf_spec = dict(f=f / downsampling, a=a_M, d=d_M)
    if shift_f == 0:                                                                        # :136-157
        fwd = back = 0.0
    elif shift_f == 1:
        idx = [i for i, fa in enumerate(mb_FA) if fa > 60]
        fwd = (f[2 * idx[0]] + f[2 * idx[-1] + 1]) / 2
    elif shift_f == 2:
        i = int(np.argmax(mb_FA))
        fwd = (f[2 * i] + f[2 * i + 1]) / 2
    else:
        raise ValueError("shift_f = %r is not an option. Options are 0,1,2" % (shift_f,))
    if shift_f:
        f = f - fwd
        back = fwd * 0.5 * (1 / dt)                                                        # kHz
    status = None
    if ftype == "ms":                                                                       # :164-165
        raise NameError("ftype 'ms': TBW is undefined in the reference (dzrf_mb.m:165)")
    elif ftype == "ap_cvx":
        b, status = mbfir.fir_ap_cvx(n, f, a, d, 1.0, Peak, opts=opts)
    elif ftype == "ap_minstopripple_cvx":
        b, status = mbfir.fir_ap_cvx(n, f, a, d, 1e4, Peak, opts=opts)
    elif ftype == "ap_minorder_cvx":                                                        # :170-177
        if min_order <= 1:
            b, status = mbfir.fir_ap(n, f, a, d, Peak, min_order, 0, 0, opts=opts, probes=probes)[:2]
        else:                                       # design with fixed n (= min_order)
            b, status = mbfir.fir_ap_cvx(int(min_order), f, a, d, 0.1, Peak, opts=opts)
    elif ftype == "ap_mintran_cvx":                                                         # :183-195
        b, status, _, f_new = mbfir.fir_ap(n, f, a, d, Peak, 0, min_tran, 0, opts=opts, probes=probes)
        b_spec["f"] = (f_new + fwd) / downsampling
        rf_spec["f"] = (f_new + fwd) / downsampling
    elif ftype == "ap_mintran_minorder_cvx":
        # the reference passes its arguments shifted by one (dzrf_mb.m:197 against fir_ap.m:1): Peak = 0.4,
        # min_order = 0.3, min_tran = 0, min_peak = dbg
        b, status = mbfir.fir_ap(n, f, a, d, 0.4, 0.3, 0, 0, opts=opts, probes=probes)[:2]
    elif ftype == "lp_minorder":
        b, status = mbfir.fir_min_order_linprog(n, f, a, d, None, opts=opts, probes=probes)
    elif ftype == "qp_cvx":
        b, status = mbfir.fir_qp_cvx(n, f, a, d, 120.0, 1e6, opts=opts)                     # :204-206
    else:
        b, status = np.zeros(0, dtype=np.complex128), None                                  # no case matches: b stays []
    if status == "Failed":                                                                  # :216-218
        return np.zeros(0, dtype=np.complex128), np.zeros(0, dtype=np.complex128), rf_spec, b_spec
    b = np.asarray(b, dtype=np.complex128).ravel()[::-1]                                    # :220
    if flip_zero:                                                                           # :223-225
        b = mbfir.fir_flip_zero(b, dbg)
    if downsampling >= 2:                                                                   # :228-231
        b = fir_upsample(b, dt, dt / downsampling)
    dt = dt / downsampling
    rf = b.copy() if ptype == "st" else mbfir.b2rf(b)                                       # :236-241
    rf_pulse = mbfir.rfscaleg(rf, dt * len(rf), gamma)                                      # :244
    t_axis = np.arange(1, len(rf) + 1) * dt                                                 # :279-280
    rf_pulse = rf_pulse * np.exp(2j * np.pi * back * t_axis)
    return rf_pulse, b, rf_spec, b_spec
