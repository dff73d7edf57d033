"""fir_flip_zero.m, host side: lower the peak amplitude of a filter by reflecting pass-band zeros about the unit
circle (which keeps |H(w)| up to a constant) and keeping the combination with the smallest peak.

Same rule as the reference: zeros more than 1 % off the unit circle are pass-band zeros (fir_flip_zero.m:35); up to
12 of them every one of the 2^Nz combinations is tried, from 13 to 19 a random 4096 of them, beyond that 4096 random
masks (:56-72); every candidate is rescaled to the DC gain of the input (:83).  The combinations are evaluated
together (one polynomial product per zero over the whole candidate set) instead of one `poly` call each.  Where the
reference draws from MATLAB's global RNG (`randperm`, `rand`), `seed` makes the draw reproducible."""
import numpy as np


def _flip(z):
    return (1.0 / np.abs(z)) * np.exp(1j * np.angle(z))          # fir_flip_zero.m:146-149


def _masks(nz, rng):
    """Columns = combinations (1 = flipped), in the reference's order (combination_2power, :153-160)."""
    if nz <= 19:
        idx = np.arange(2 ** nz)
        if nz > 12:
            idx = np.sort(rng.permutation(2 ** nz)[:2 ** 12])
        # column c of combination_2power(n): row r is 1 - bit (n-1-r) of c
        rows = np.arange(nz)[:, None]
        return 1 - ((idx[None, :] >> (nz - 1 - rows)) & 1)
    return np.round(rng.random((nz, 2 ** 12))).astype(np.int64)    # combination_MC, :162-169


def fir_flip_zero(h, dbg=0, *, seed=None, return_info=False):
    """`h_new = fir_flip_zero(h, dbg)`: n taps in, n taps out (complex)."""
    h = np.asarray(h, dtype=np.complex128).ravel()
    N = len(h)
    Z = np.roots(h)
    pb = np.nonzero((np.abs(Z) > 1 + 1e-2) | (np.abs(Z) < 1 - 1e-2))[0]
    nz = len(pb)
    if nz == 0:
        return (h.copy(), dict(n_passband_zeros=0, candidates=1)) if return_info else h.copy()
    rng = np.random.default_rng(seed)
    mask = _masks(nz, rng)                                          # nz x Num
    num = mask.shape[1]
    zsel = np.where(mask == 1, _flip(Z[pb])[:, None], Z[pb][:, None])          # pass-band zeros per candidate
    coef = np.zeros((num, N), dtype=np.complex128)
    coef[:, 0] = 1.0
    deg = 0
    fixed = np.ones(len(Z), dtype=bool)
    fixed[pb] = False
    for r in Z[fixed]:                                              # the stop-band zeros are common to all candidates
        coef[:, 1:deg + 2] -= r * coef[:, :deg + 1]
        deg += 1
    for j in range(nz):
        coef[:, 1:deg + 2] -= zsel[j][:, None] * coef[:, :deg + 1]
        deg += 1
    coef *= (np.sum(h) / np.sum(coef, axis=1))[:, None]              # :83
    peak = np.max(np.abs(coef), axis=1)
    best = int(np.argmin(peak))                                      # :102 (first minimum, as MATLAB's min)
    if dbg >= 1:
        print("reduce peak amplitude from %6.4f to %6.4f by %6.4f" % (np.abs(h).max(), peak[best], (np.abs(h).max() - peak[best]) / np.abs(h).max()))
    if return_info:
        return coef[best], dict(n_passband_zeros=nz, candidates=num, peak_before=float(np.abs(h).max()), peak_after=float(peak[best]),
                                mask=mask[:, best].copy())
    return coef[best]
