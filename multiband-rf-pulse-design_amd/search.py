"""Outer searches around the convex designers: host-side mirror of fir_ap.m.

fir_ap(n, f, a, d, Peak, min_order, min_tran, ...) reproduces the reference's two bisections
(fir_ap.m:63-106 transition widening, :143-176 order) probe for probe when `probes == 1`.  With
`probes > 1` every round hands that many interior points to mbfir.solve_batch at once (one HIP stream
each), which shrinks the bracket by (probes + 1) per round instead of 2; for a feasibility predicate
that is monotone in the searched parameter -- what a bisection assumes anyway -- the bracket ends at
the same threshold.
"""
import math

import numpy as np

DF_THRE = 0.0005        # fir_ap.m:47
LAMBDA = 0.1            # fir_ap.m:46 (minimise total energy instead of the stop-band ripple)


def _widen(f, f_add):
    fn = np.array(f, dtype=np.float64).copy()
    fn[0::2] -= f_add
    fn[1::2] += f_add
    return fn


def fir_ap(n, f, a, d, Peak=1e-3, min_order=0, min_tran=0, min_peak=0, dbg=0, *, probes=1, opts=None, log=None):
    """Returns (h, status, n_op, f_op) like fir_ap.m.  probes: candidates evaluated concurrently per
    search round (1 = the reference's bisection).  log: optional list receiving (kind, value, status)."""
    import mbfir
    if n is None or f is None or a is None or d is None:
        raise ValueError("not enough input")
    if min_peak:
        raise NotImplementedError("min_peak (fir_flip_zero.m) is outside the built path")
    if not 0 <= min_tran <= 1:
        raise ValueError("invalid input of min_tran")
    if not 0 <= min_order <= 1:
        raise ValueError("invalid input of min_order")
    f = np.asarray(f, dtype=np.float64).ravel()
    note = (lambda *t: log.append(t)) if log is not None else (lambda *t: None)

    def design(jobs):
        """jobs: list of (n, f); returns [(h, status)] -- one call for one job, a batch otherwise."""
        if len(jobs) == 1:
            nn, ff = jobs[0]
            return [mbfir.fir_ap_cvx(nn, ff, a, d, LAMBDA, Peak, opts=opts)]
        return mbfir.solve_batch([("fir_ap_cvx", (nn, ff, a, d, LAMBDA, Peak)) for nn, ff in jobs], opts=opts,
                                 streams=min(len(jobs), 4))

    n_op, f_op = n, f
    h1, status1 = design([(n, f)])[0]                                   # fir_ap.m:51
    if status1 == "Failed":
        raise ValueError("original parameters are too tight")          # fir_ap.m:52-54
    h, status = h1, status1
    if min_tran == 0 and min_order == 0:
        return h, status, n_op, f_op

    if min_tran > 0:                                                   # fir_ap.m:63-106
        df_min = float(np.min(f[2::2] - f[1:-1:2]))
        bot, top = 0.0, df_min / 2
        while True:
            mids = [bot + (top - bot) * (q + 1) / (probes + 1) for q in range(probes)]
            res = design([(n, _widen(f, m)) for m in mids])
            new_bot, new_top = bot, top
            for m, (h0, s0) in zip(mids, res):
                note("f_add", m, s0)
                if s0 == "Failed":
                    new_top = min(new_top, m)
                    break                                               # everything above is taken as failed too
                h, status, new_bot = h0, s0, m
            bot, top = new_bot, new_top
            if top - bot < DF_THRE:
                break
        f_add = bot * min_tran                                          # fir_ap.m:110-131
        f_new = _widen(f, f_add)
        h0, s0 = design([(n, f_new)])[0]
        note("f_add_final", f_add, s0)
        if s0 == "Failed":
            f_new = _widen(f, bot)
        else:
            h, status = h0, s0
        f = f_new
        f_op = f_new

    if min_order > 0:                                                  # fir_ap.m:143-176
        n_top, n_bot = n, 2
        while n_top - n_bot > 1:
            if probes == 1:
                mids = [int(math.ceil((n_top + n_bot) / 2))]
            else:
                mids = sorted({n_bot + int(math.ceil((n_top - n_bot) * (q + 1) / (probes + 1))) for q in range(probes)}
                              - {n_bot, n_top})
                if not mids:
                    mids = [int(math.ceil((n_top + n_bot) / 2))]
            res = design([(m, f) for m in mids])
            for m, (h0, s0) in sorted(zip(mids, res), reverse=True):   # from the longest filter down
                note("n", m, s0)
                if s0 == "Failed":
                    n_bot = max(n_bot, m)
                    break                                               # everything shorter is taken as failed too
                h, status, n_top = h0, s0, m
        if min_order == 1:
            n_op = n_top
        else:
            n_op = int(math.ceil(n * (1 - min_order) + n_top * min_order))
            h, status = design([(n_op, f)])[0]
            note("n_final", n_op, status)
    return h, status, n_op, f_op
