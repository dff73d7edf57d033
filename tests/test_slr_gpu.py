"""Device inverse SLR (mbfir_b2a / mbfir_ab2rf / mbfir_b2rf) against the oracle and the reference-C fixture."""
import json
import os

import numpy as np
import pytest

import mbfir
from oracle import slr

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "slr_golden.json")))


def cx(d):
    return np.array(d["re"]) + 1j * np.array(d["im"])


@pytest.mark.parametrize("name", sorted(GOLD))
def test_device_b2a_against_oracle_and_c_twin(name):
    g = GOLD[name]
    b = cx(g["b"])
    a = mbfir.b2a(b)
    assert np.max(np.abs(a - slr.b2a(b))) <= 1e-12            # same algorithm: rounding only
    assert np.max(np.abs(a - np.conj(cx(g["a_c"])))) <= g["tol_a"] + 1e-12


@pytest.mark.parametrize("name", sorted(GOLD))
def test_device_ab2rf_against_cabc2rf(name):
    g = GOLD[name]
    rf = mbfir.ab2rf(np.conj(cx(g["a_c"])), cx(g["b"]))
    assert np.max(np.abs(rf - cx(g["rf_c"]))) <= 1e-12


@pytest.mark.parametrize("name", sorted(GOLD))
def test_device_b2rf_against_the_c_chain(name):
    g = GOLD[name]
    rf = mbfir.b2rf(cx(g["b"]))
    assert np.max(np.abs(rf - cx(g["rf_c"]))) <= 2 * g["tol_a"] + 1e-11


@pytest.mark.parametrize("n", [2, 63, 64, 65, 300, 1000, 2048])
def test_device_b2rf_sizes(n):
    rng = np.random.default_rng(n)
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * np.hanning(n + 2)[1:-1]
    b = b * 0.8 / np.max(np.abs(np.fft.fft(b, 16 * n)))
    a, rf = mbfir.b2a(b), mbfir.b2rf(b)
    ao = slr.b2a(b)
    assert np.max(np.abs(a - ao)) <= 1e-11
    assert np.max(np.abs(rf - slr.ab2rf(ao, b))) <= 1e-9
    assert np.max(np.abs(mbfir.ab2rf(ao, b) - slr.ab2rf(ao, b))) <= 1e-11


def test_device_b2a_clipping_branch():
    rng = np.random.default_rng(5)
    b = rng.standard_normal(24) * np.hanning(24)
    assert np.max(np.abs(mbfir.b2a(b) - slr.b2a(b))) <= 1e-9


def test_designed_beta_to_rf_pulse():
    """fir_ap_cvx -> b2rf -> rfscaleg, the tail of dzrf_mb.m:206-244, against the oracle on the same taps."""
    f, a, d = mbfir.spec.spec_c13_bssfp(64)
    h, status = mbfir.fir_ap_cvx(64, f, a, d, 0.0, 1e-3)[:2]
    assert status == "Solved"
    rf = mbfir.rfscaleg(mbfir.b2rf(h), 64 * 0.04, 1.0705)
    ref = slr.rfscaleg(slr.b2rf(h), 64 * 0.04, 1.0705)
    assert np.max(np.abs(rf - ref)) <= 1e-9 * np.max(np.abs(ref))


def test_ab2rf_rejects_long_and_ragged_inputs():
    with pytest.raises(mbfir.MbfirError):
        mbfir.b2rf(np.ones(2049) * 1e-4)
    with pytest.raises(ValueError):
        mbfir.ab2rf(np.ones(4), np.ones(5))


def test_device_chain_on_the_reference_data_file():
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "slr_newmat.json")))
    h, rfm = cx(g["h"]), cx(g["rfm"])
    got = mbfir.b2rf(h)
    assert np.max(np.abs(got - rfm)) / np.max(np.abs(rfm)) < 5e-3
    assert np.max(np.abs(got - slr.b2rf(h))) / np.max(np.abs(rfm)) < 1e-6        # alpha -> 0 amplifies rounding
