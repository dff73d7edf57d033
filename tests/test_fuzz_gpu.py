"""Seeded random band specs through all four designers: device (default path) against the CPU oracle -- same verdict,
taps within 1e-6 relative l-inf whenever both solves are clean (not the reduced-accuracy exit)."""
import warnings

import numpy as np
import pytest
from conftest import relinf

import mbfir
from oracle import conic_ipm, designers

pytestmark = pytest.mark.gpu


def random_bands(rng, lo, hi, kmax):
    k = int(rng.integers(2, kmax + 1))
    widths = rng.uniform(0.03, 0.12, k) * (hi - lo) / 2
    gaps = rng.uniform(0.06, 0.2, k + 1) * (hi - lo) / 2
    scale = (hi - lo) / (widths.sum() + gaps.sum())
    edges, x = [], lo + gaps[0] * scale
    for w, g in zip(widths, gaps[1:]):
        edges += [x, x + w * scale]
        x += (w + g) * scale
    return np.array(edges), k


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    which = ["fir_ap_cvx", "fir_qp_cvx", "fir_linprog", "fir_qprog_phs"][seed % 4]
    n = int(rng.integers(16, 56))
    if which == "fir_linprog" and rng.random() < 0.5:
        f, k = random_bands(rng, 0.0, 1.0, 3)
    else:
        f, k = random_bands(rng, -1.0, 1.0, 4)
    amp = np.where(rng.random(k) < 0.5, 0.0, rng.uniform(0.3, 1.0, k))
    if not np.any(amp > 0):
        amp[int(rng.integers(0, k))] = 0.8
    a = np.repeat(amp, 2)
    d = rng.uniform(0.01, 0.05, k)
    if which == "fir_ap_cvx":
        return which, (n, f, a * 0.3, d * 0.3, 0.1, 10 ** rng.uniform(-2.5, -1))
    if which == "fir_qp_cvx":
        return which, (n, f, a, d, float(rng.uniform(2, 20)), [10.0, [0.1, 5.0]][seed // 4 % 2])
    if which == "fir_linprog":
        return which, (n, f, a, d)
    ph = np.repeat(np.exp(1j * rng.uniform(-0.3, 0.3, k)), 2)
    dc = np.where(amp > 0, d * np.exp(0.3j), d.astype(complex))       # pass bands carry a phase ripple (fir_qprog_phs.m)
    return which, (n, f, a * ph, dc)


@pytest.mark.parametrize("seed", range(64))
def test_random_spec_matches_the_oracle(seed):
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    which, args = make_case(seed)
    hg, sg, ig = getattr(mbfir, which)(*args, info=True)
    ho, so, io = getattr(designers, which)(*args, info=True)
    assert sg == so, (which, args[0], ig["rc"], io["status"])
    if sg == "Solved":
        clean = io["status"] == conic_ipm.STATUS_OPTIMAL and ig["relgap"] <= 1e-6
        assert abs(ig["pcost"] - io["pcost"]) <= 1e-6 * max(1.0, abs(io["pcost"]))
        if clean:
            # absolute floor: a spec whose optimum is h ~ 1e-10 has nothing to compare relatively
            assert np.max(np.abs(hg - ho)) <= 1e-6 * max(np.max(np.abs(ho)), 1e-3), (which, args[0], ig["iters"], io["iters"])


@pytest.mark.parametrize("seed", range(12))
def test_random_quadratic_phase_specs_with_large_peak_weights(seed):
    """fir_qp_cvx the way dzrf_mb calls it (k ~ 100, obj ~ 1e6: dzrf_mb.m:210-213) on random band specs: the regime in
    which the error and tap-peak cones become nearly active with NT weights far above the rest and the extended-precision
    KKT solve takes over.  Device and oracle must agree on the verdict, the objective and -- when both solves
    are clean -- the taps."""
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    rng = np.random.default_rng(4000 + seed)
    n = int(rng.integers(40, 120))
    f, k = random_bands(rng, -1.0, 1.0, 3)
    f = f * rng.uniform(0.15, 0.5)                               # narrow bands around DC, as the spectral pulses have
    amp = np.where(rng.random(k) < 0.4, 0.0, rng.uniform(0.5, 0.9, k))
    amp[int(rng.integers(0, k))] = 0.8
    a, d = np.repeat(amp, 2), rng.uniform(0.01, 0.05, k)
    kq, obj = float(rng.uniform(40, 140)), float(10 ** rng.uniform(4, 6.5))
    grid = int(rng.choice([0, 4 * n, 8 * n]))
    hg, sg, ig = mbfir.fir_qp_cvx(n, f, a, d, kq, obj, opts=mbfir.make_opts(grid_m=grid), info=True)
    ho, so, io = designers.fir_qp_cvx(n, f, a, d, kq, obj, grid_m=grid, info=True)
    assert sg == so, (n, grid, ig["rc"], io["status"], ig["relgap"], io.get("relgap"))
    if sg == "Solved":
        assert abs(ig["pcost"] - io["pcost"]) <= 1e-6 * max(1.0, abs(io["pcost"]))
        if io["status"] == conic_ipm.STATUS_OPTIMAL and ig["relgap"] <= 1e-8:
            assert np.max(np.abs(hg - ho)) <= 1e-4 * max(np.max(np.abs(ho)), 1e-3)      # E + obj Peak is flat around its minimiser


@pytest.mark.parametrize("mode", ["edges", "orders", "dd"])
def test_random_lock_step_units_equal_their_single_solves(mode):
    """tools/gpu_fuzz_lockstep.py inside the suite (round 5): 40 seeds x 6 variants per mode through one mbfir_solve_batch call per four
    seeds on three contexts -- heterogeneous units (band edges; orders too; the extended-precision solve on) -- every job against its
    single-design solve, bit for bit.  The suite's fixed cases had passed while a missing store drain in a fused kernel let 1 job in 720
    differ from run to run; this many units in flight is what shows such a thing."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = {"edges": ["0", "40", "4", "6", "edges"], "orders": ["0", "40", "4", "6", "orders"], "dd": ["0", "40", "4", "6", "same", "dd"]}[mode]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_fuzz_lockstep.py")] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("lock-step fuzz")]
    assert last and " 240 jobs, 0 mismatches" in last[-1], r.stdout[-1500:]


def test_batches_reproduce_themselves_run_to_run():
    """tools/gpu_determinism.py inside the suite: the bench's batch (64 headline designs, units of 16 on 4
    streams), BASELINE config 3 in lock-step units and a heterogeneous batch, each solved three times -- every repetition must
    reproduce the first bit for bit (verdicts, iterations, objective, taps): nothing in a solve may depend on the order in which
    workgroups or streams happen to run."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_determinism.py"), "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if "repetitions:" in ln]
    assert len(lines) == 3 and all(ln.rstrip().endswith(" 0 results differ from the first run") for ln in lines), r.stdout[-1500:]
