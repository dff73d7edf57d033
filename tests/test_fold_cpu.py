"""Host-side analysis of the frequency grid for the lattice kernels (mbfir_test_fold; no GPU needed): the +w / -w pairing
of a symmetric grid and the cut into equally spaced runs."""
import numpy as np
import pytest

import mbfir
from oracle.assemble import matlab_linspace


def test_symmetric_grid_folds_to_half_the_entries():
    r = mbfir.test_fold(matlab_linspace(-np.pi, np.pi, 16384))
    assert r == dict(ok=1, nfold=8192, pairs=8192, runs=128, longest=64, bad=0)
    u = mbfir.test_fold(matlab_linspace(-np.pi, np.pi, 16384), fold=False)
    assert u == dict(ok=1, nfold=16384, pairs=0, runs=256, longest=64, bad=0)


def test_odd_grid_keeps_zero_on_its_own():
    r = mbfir.test_fold(matlab_linspace(-np.pi, np.pi, 1201))
    assert r["ok"] == 1 and r["nfold"] == 601 and r["pairs"] == 600 and r["bad"] == 0


def test_one_sided_grid_has_no_partners():
    r = mbfir.test_fold(matlab_linspace(0, np.pi, 960))
    assert r["ok"] == 1 and r["nfold"] == 960 and r["pairs"] == 0 and r["runs"] == 15 and r["bad"] == 0


def test_band_edges_stay_single_and_cut_the_runs():
    edges = np.pi * np.array([-0.6, -0.35, -0.1, 0.15, 0.45, 0.8])
    w = np.concatenate([matlab_linspace(-np.pi, np.pi, 996), edges])           # fir_ap_cvx.m:47-48 (the solver gets them reordered)
    rng = np.random.default_rng(5)
    for perm in (np.arange(len(w)), rng.permutation(len(w))):
        r = mbfir.test_fold(w[perm])
        assert r["ok"] == 1 and r["bad"] == 0 and r["pairs"] == 498 and r["nfold"] == 498 + 6
        assert 498 / 64 <= r["runs"] <= 8 + 2 * 6 + 6


def test_symmetric_edges_pair_up_and_duplicates_do_not():
    w = np.concatenate([matlab_linspace(-np.pi, np.pi, 400), [0.3, -0.3, 0.7, 0.7]])
    r = mbfir.test_fold(w)
    assert r["bad"] == 0 and r["pairs"] == 200 + 1 and r["nfold"] == 200 + 1 + 2       # (+0.3, -0.3) pair; the two 0.7 stay apart


def test_irregular_grid_is_refused():
    r = mbfir.test_fold(np.random.default_rng(1).uniform(-3, 3, 500))
    assert r["ok"] == 0 and r["bad"] == 0


@pytest.mark.parametrize("m", [7, 64, 65, 1000, 4099])
def test_every_frequency_lands_in_exactly_one_entry(m):
    for lo in (-np.pi, 0.0, -1.0):
        r = mbfir.test_fold(matlab_linspace(lo, np.pi, m))
        assert r["bad"] == 0 and r["nfold"] + r["pairs"] == m
