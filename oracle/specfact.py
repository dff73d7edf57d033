"""Oracle post-processing (test infrastructure, see oracle/__init__.py).

x -> autocorrelation r -> minimum-phase taps, restating fir_ap_cvx.m:185-186
(reshape), :264-284 (fmp2) and :294-304 (mag2mp) with numpy FFTs.
"""
import numpy as np


def x_to_r(x, n):
    """fir_ap_cvx.m:185-186: two-sided Hermitian autocorrelation, length 2n-1."""
    x = np.asarray(x, dtype=np.float64)
    r = np.concatenate([[x[0] + 0j], x[1:n] + 1j * x[n:2 * n - 1]])
    return np.concatenate([np.conj(r[:0:-1]), r])


def mag2mp(x):
    """fir_ap_cvx.m:294-304."""
    n = len(x)
    xl = np.log(x)
    xlf = np.fft.fft(xl)
    xlfp = np.zeros(n, dtype=np.complex128)
    xlfp[0] = xlf[0]                              # keep DC
    xlfp[1:n // 2] = 2 * xlf[1:n // 2]            # double positive freqs
    xlfp[n // 2] = xlf[n // 2]                    # keep Nyquist
    xlaf = np.fft.ifft(xlfp)                      # negative freqs zeroed
    return np.exp(xlaf)


def fmp2(r):
    """fir_ap_cvx.m:264-284: spectral factorisation of an odd-length
    autocorrelation; returns (l+1)/2 taps as a row vector."""
    r = np.asarray(r, dtype=np.complex128).ravel()
    l = len(r)
    if l % 2 == 0:
        raise ValueError("filter length must be odd")
    # :272  lp = 8*exp(ceil(log(l)/log(2))*log(2)); the ceil/floor split of the
    # padding below makes the padded length the exact power of two.
    lp = 8 * (1 << int(np.ceil(np.log(l) / np.log(2))))
    pad = lp - l
    hp = np.concatenate([np.zeros(-(-pad // 2)), r, np.zeros(pad // 2)])   # :273
    hpf = np.fft.fftshift(np.fft.fft(np.fft.fftshift(hp)))                  # fftc :274
    hpfmp = mag2mp(np.sqrt(np.abs(hpf)))                                    # :281
    hpmp = np.fft.ifft(np.fft.fftshift(np.conj(hpfmp)))                     # :282
    return hpmp[: (l + 1) // 2]                                             # :283
