import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # the oracle's BLAS calls are small (n <= 200 in the CPU suite): more than four threads only oversubscribe the cores the
    # C++ twin's OpenMP team and the test workers share (8 cores here: 277 s with the default team, 62 s with four threads)
    try:
        from threadpoolctl import threadpool_limits
        config._mbfir_blas_limit = threadpool_limits(limits=min(4, os.cpu_count() or 1), user_api="blas")
    except Exception:                                       # noqa: BLE001  (threadpoolctl missing: the defaults stay)
        pass


# ---- shared specs ---------------------------------------------------------
# S-C13: bSSFP C-13 five-band lactate spec of bSSFP_pulse_sb_mb.m:9-52 evaluated at n=100
F100 = [-0.241994, -0.233994, -0.152431, -0.144431, -0.083851, -0.075851, -0.052798, -0.044798, -0.004, 0.004]
A_C13 = [0.0] * 8 + [0.500125, 0.500125]
D_C13 = [0.00250001] * 4 + [0.00866503]


def c13(n, regime="minorder"):
    """S-C13 band edges: regime (i) min-order keeps f, regime (ii) fixed duration scales f by 100/n."""
    s = 1.0 if regime == "minorder" else 100.0 / n
    return [x * s for x in F100], list(A_C13), list(D_C13)


CASES = {
    # name: (designer, args)      -- small enough for the oracle to finish in about a second
    "ap_lowpass20": ("fir_ap_cvx", (20, [-0.2, 0.2, 0.5, 1.0], [1, 1, 0, 0], [0.05, 0.02], 1.0, 1e-1)),
    "ap_twoband33": ("fir_ap_cvx", (33, [-0.6, -0.35, -0.1, 0.15, 0.45, 0.8], [0, 0, 0.7, 0.7, 0, 0], [0.01, 0.02, 0.01], 0.1, 1e-2)),
    "ap_c13_58": ("fir_ap_cvx", (58,) + tuple(c13(58)) + (0.1, 1e-3)),       # spike cones active; 'min_order = 58'
    "ap_c13_64": ("fir_ap_cvx", (64,) + tuple(c13(64)) + (0.1, 1e-3)),
    "qp_modelB25": ("fir_qp_cvx", (25, [-0.5, -0.3, -0.1, 0.1, 0.3, 0.5], [0, 0, 1, 1, 0, 0], [0.01, 0.02, 0.01], 20.0, [0.1, 5.0])),
    "qp_modelA48": ("fir_qp_cvx", (48, [-0.5, -0.3, -0.1, 0.1, 0.3, 0.5], [0, 0, 1, 1, 0, 0], [0.05, 0.05, 0.05], 5.0, 10.0)),
    "lin_real64": ("fir_linprog", (64, [0, 0.2, 0.3, 1], [1, 1, 0, 0], [0.01, 0.01])),
    "lin_real33": ("fir_linprog", (33, [0, 0.25, 0.45, 1], [1, 1, 0, 0], [0.02, 0.02])),
    "lin_cplx31": ("fir_linprog", (31, [-1, -0.4, -0.2, 0.3, 0.5, 1], [0, 0, 1, 0.8, 0, 0], [0.01, 0.02, 0.01])),
    "lin_cplx32": ("fir_linprog", (32, [-1, -0.4, -0.2, 0.3, 0.5, 1], [0, 0, 1, 0.8, 0, 0], [0.01, 0.02, 0.01])),
    "qphs21": ("fir_qprog_phs", (21, [-0.6, -0.3, -0.1, 0.1, 0.3, 0.6], [0, 0, 1, 1, 0, 0], [0.02, 0.05 * np.exp(0.3j), 0.02])),
    "qphs22": ("fir_qprog_phs", (22, [-0.6, -0.3, -0.1, 0.1, 0.3, 0.6], [0, 0, 1, 1, 0, 0], [0.02, 0.05 * np.exp(0.3j), 0.02])),
    # infeasible instances: the bisection wrappers (fir_ap.m:86-93) rely on a definite 'Failed'
    "ap_c13_50_infeasible": ("fir_ap_cvx", (50,) + tuple(c13(50)) + (0.1, 1e-3)),
    "lin_real31_infeasible": ("fir_linprog", (31, [0, 0.2, 0.3, 1], [1, 1, 0, 0], [0.01, 0.01])),
    "qp_modelA24_infeasible": ("fir_qp_cvx", (24, [-0.5, -0.3, -0.1, 0.1, 0.3, 0.5], [0, 0, 1, 1, 0, 0], [0.01, 0.02, 0.01], 20.0, 100.0)),
}
WHICH = {"fir_ap_cvx": 0, "fir_qp_cvx": 1, "fir_linprog": 2, "fir_qprog_phs": 3}


def relinf(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        return json.load(fh)
