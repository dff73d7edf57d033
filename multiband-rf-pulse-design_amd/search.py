"""Outer searches around the convex designers: host-side mirrors of fir_ap.m, fir_qp.m,
ss/fir_min_order_linprog.m and ss/fir_min_order_qprog_phs.m.

fir_ap(n, f, a, d, Peak, min_order, min_tran, ...) reproduces the reference's two bisections
(fir_ap.m:63-106 transition widening, :143-176 order) probe for probe when `probes == 1`.  With
`probes > 1` every round hands that many interior points to mbfir.solve_batch at once -- the probes of a
transition-width round (one order, different band edges) as ONE lock-step unit, the probes of an order round one per
HIP stream --, which shrinks the bracket by (probes + 1) per round instead of 2; for a feasibility predicate
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


UNIT_ORDER_MAX = 200      # longest filter whose min-order probes share lock-step units (see fir_ap's design())


def fir_ap(n, f, a, d, Peak=1e-3, min_order=0, min_tran=0, min_peak=0, dbg=0, *, probes=1, opts=None, log=None, unit_probes=True):
    """Returns (h, status, n_op, f_op) like fir_ap.m.  probes: candidates evaluated concurrently per
    search round (1 = the reference's bisection).  log: optional list receiving (kind, value, status).
    unit_probes: the probes of a round that share the order run as one lock-step unit (False: one design per stream)."""
    import mbfir
    if n is None or f is None or a is None or d is None:
        raise ValueError("not enough input")
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
        batch = [("fir_ap_cvx", (nn, ff, a, d, LAMBDA, Peak)) for nn, ff in jobs]
        one_order = len({nn for nn, _ in jobs}) == 1
        if unit_probes and (one_order or max(nn for nn, _ in jobs) <= UNIT_ORDER_MAX):
            # the probes of a round -- one order and different band edges (transition-width round), or different orders (min-order
            # round; round 5) -- as ONE lock-step unit per size bucket: every launch carries all of them (heterogeneous unit,
            # DESIGN.md section 5) instead of one design per stream.  A unit of different orders runs every lane at the size of its
            # largest and ends with its slowest: measured on S-C13 (tools/gpu_search_orders.py) it wins at n = 100 (0.098 s against
            # 0.119 s one design per stream, 0.142 s bisection) and loses at n = 512 (2.1 s against 1.5 s), so long filters keep the streams
            return mbfir.solve_batch(batch, opts=mbfir.opts_with(opts, lanes=len(batch)), streams=1 if one_order else min(len(jobs), 2))
        return mbfir.solve_batch(batch, opts=opts, streams=min(len(jobs), 4))

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
    if min_peak and len(h):                                            # fir_ap.m:199-211 (skipped by the early return above)
        h = mbfir.fir_flip_zero(h, dbg)
    return h, status, n_op, f_op


QP_DF_THRE = 0.001      # fir_qp.m:30
QP_LAMBDA = 1e5         # fir_qp.m:31


def fir_qp(n, f, a, d, min_order=0, min_tran=0, min_peak=0, dbg=0, *, opts=None, log=None, designer=None):
    """Returns (h, status) like fir_qp.m: the low-pass transition-width and order bisections of that file,
    which (as in the reference, fir_qp.m:46,69,97,111,131) run on fir_ap_cvx(n, f, a, d, 1e5) with the default
    Peak.  f has four edges [f1 fp fs f4]; the widened spec is [-fp fp fs f4] (fir_qp.m:68).
    designer: callable (n, f) -> (h, status) replacing the device call (tests)."""
    import mbfir
    if n is None or f is None or a is None or d is None:
        raise ValueError("not enough input")                                          # fir_qp.m:33
    f = np.asarray(f, dtype=np.float64).ravel()
    note = (lambda *t: log.append(t)) if log is not None else (lambda *t: None)
    design = designer or (lambda nn, ff: mbfir.fir_ap_cvx(nn, ff, a, d, QP_LAMBDA, opts=opts))
    h1, status1 = design(n, f)                                                        # fir_qp.m:46
    if status1 == "Failed":
        raise ValueError("original parameters are too tight")                         # fir_qp.m:47-49
    h, status = h1, status1
    df_top = None
    if min_tran > 0:                                                                  # fir_qp.m:56-85
        centre = (f[2] + f[1]) / 2
        df_top, df_bot = (f[2] - f[1]) / 2, 0.0
        while True:
            df_mid = (df_top + df_bot) / 2
            f_new = np.array([-(centre - df_mid), centre - df_mid, centre + df_mid, f[3]])
            h0, s0 = design(n, f_new)
            note("df", df_mid, s0)
            if s0 == "Failed":
                df_bot = df_mid
            else:
                h, status, df_top = h0, s0, df_mid
            if df_top - df_bot < QP_DF_THRE:
                break
    if min_tran != 0:                                                                 # fir_qp.m:88-100
        if not 0 < min_tran <= 1:
            raise ValueError("invalid input of min_tran")
        centre = (f[2] + f[1]) / 2
        df_new = ((f[2] - f[1]) / 2) * (1 - min_tran) + df_top * min_tran
        f = np.array([-(centre - df_new), centre - df_new, centre + df_new, f[3]])
        h, status = design(n, f)
        note("df_final", df_new, status)
    n_top = n
    if min_order > 0:                                                                 # fir_qp.m:104-124
        n_bot = 2
        while True:
            n_mid = int(math.ceil((n_top + n_bot) / 2))
            h0, s0 = design(n_mid, f)
            note("n", n_mid, s0)
            if s0 == "Failed":
                n_bot = n_mid
            else:
                h, status, n_top = h0, s0, n_mid
            if n_top - n_bot == 1:
                break
    if min_order not in (0, 1):                                                       # fir_qp.m:126-136
        if not 0 < min_order < 1:
            raise ValueError("invalid input of min_order")
        n_new = int(math.ceil(n * (1 - min_order) + n_top * min_order))
        h, status = design(n_new, f)
        note("n_final", n_new, status)
    if min_peak and len(h):                                                           # fir_qp.m:139-151
        h = mbfir.fir_flip_zero(h, dbg)
    return h, status


def _min_order(which, n, f, a, d, even_odd, probes, opts, log, designer):
    """ss/fir_min_order_linprog.m:56-233 / ss/fir_min_order_qprog_phs.m:40-213 (the two files differ only in
    the designer they call): bisection over the half length, odd lengths first, then even lengths below the
    best odd one; the shorter of the two wins, odd on a tie (:222-229)."""
    import mbfir
    note = (lambda *t: log.append(t)) if log is not None else (lambda *t: None)
    if even_odd not in (1, 2):
        even_odd = 0

    def design(taps):
        if designer is not None:
            return [designer(t) for t in taps]
        if len(taps) == 1:
            return [getattr(mbfir, which)(taps[0], f, a, d, opts=opts)]
        return mbfir.solve_batch([(which, (t, f, a, d)) for t in taps], opts=opts, streams=min(len(taps), 4))

    def search(n_top, taps_of):
        """The reference's loop (:79-140): the first probe is n_top itself."""
        best = None
        n_bot, n_cur = 1, n_top
        while n_top - n_bot > 1:
            if probes == 1 or n_cur == n_top:
                cand = [n_cur]
            else:                                        # several interior points of (n_bot, n_top) at once
                cand = sorted({n_bot + int(math.ceil((n_top - n_bot) * (q + 1) / (probes + 1))) for q in range(probes)}
                              - {n_bot, n_top}) or [n_cur]
            res = design([taps_of(c) for c in cand])
            hit_fail = False
            for c, (h0, s0) in sorted(zip(cand, res), reverse=True):     # from the longest filter down
                note("n", taps_of(c), s0)
                if s0 == "Solved":
                    best, n_top = h0, c
                else:
                    n_bot = max(n_bot, c)
                    hit_fail = True
                    break                                # everything shorter is taken as failed too
            n_cur = n_bot if (not hit_fail and n_top == n_bot + 1) else int(math.ceil((n_top + n_bot) / 2))
        return best

    n_odd_max = 2 * ((n - 1) // 2) + 1
    n_even_max = 2 * (n // 2)
    best_odd = best_even = None
    if even_odd != 2:
        best_odd = search((n_odd_max + 1) // 2, lambda c: 2 * c - 1)
    if even_odd != 1:
        top = n_even_max // 2 if best_odd is None else min(n_even_max // 2, (len(best_odd) + 1) // 2)
        best_even = search(top, lambda c: 2 * c)
    if best_odd is None and best_even is None:
        return np.zeros(0, dtype=np.complex128), "Failed"
    if best_odd is None:
        return best_even, "Solved"
    if best_even is None:
        return best_odd, "Solved"
    return (best_odd if len(best_odd) < len(best_even) else best_even), "Solved"


def fir_min_order_linprog(n, f, a, d, even_odd=0, dbg=0, *, probes=1, opts=None, log=None, designer=None):
    """`[h, status] = fir_min_order_linprog(n, f, a, d, even_odd, dbg)` (ss/fir_min_order_linprog.m:56):
    shortest linear-phase filter of at most n taps; even_odd 1 = odd lengths only, 2 = even only."""
    if n is None or f is None or a is None or d is None:
        raise ValueError("Usage: function [h, status] = fir_min_order(n, f, a, d, even_odd, dbg)")
    return _min_order("fir_linprog", n, f, a, d, even_odd, probes, opts, log, designer)


def fir_min_order_qprog_phs(n, f, a, d, even_odd=0, dbg=0, *, probes=1, opts=None, log=None, designer=None):
    """`[h, status] = fir_min_order_qprog_phs(n, f, a, d, even_odd, dbg)` (ss/fir_min_order_qprog_phs.m:40)."""
    if n is None or f is None or a is None or d is None:
        raise ValueError("Usage: function [h, status] = fir_min_order(n, f, a, d, even_odd, dbg)")
    return _min_order("fir_qprog_phs", n, f, a, d, even_odd, probes, opts, log, designer)
