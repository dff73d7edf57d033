"""Generate tests/golden/slr_golden.json -- inverse-SLR vectors from the REFERENCE'S OWN compiled C.

Needs oracle/_ref/libslr_ref.so (`make -C oracle`, which compiles rf_tools/mex5/{b2a.code.c, cabc2rf.code.c,
four1.c} where they lie under /root/reference; nothing of them is copied here).  Per case:
  b       input beta polynomial: taps of a committed golden design (tests/golden/golden.json) or a seeded
          random/windowed-sinc polynomial, scaled to a physical max|B(w)| (sin(flip/2): 0.26, 0.7071, 0.966)
  a_c     b2a.code.c(b)                                       -- reference output, verbatim
  rf_c    cabc2rf.code.c(conj(a_c), b)                        -- reference output, verbatim; conj(a_c) is the
          alpha on MATLAB's frequency axis (oracle/slr.py header explains the conjugate)
  tol_a   agreement the two paddings (8 n vs power of two) allow for b2a at that max|B|
slr_newmat.json is the reference's own data file rf_tools/mex5/new.mat converted to JSON (scipy.io.loadmat; variables
h, hn, rf, rfm), nothing computed: the only stored output of the reference's b2rf chain.  Its beta has max|B| = 1.0014,
i.e. it sits on the clipping branch where alpha touches zero, so it pins the chain to 1e-2 only.
Run:  python tests/golden/make_golden_slr.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import slr  # noqa: E402


def cplx(v):
    v = np.asarray(v, dtype=np.complex128)
    return {"re": [float(t) for t in v.real], "im": [float(t) for t in v.imag]}


def scaled(h, peak):
    h = np.asarray(h, dtype=np.complex128)
    return h * (peak / np.max(np.abs(np.fft.fft(h, 64 * len(h)))))


def main():
    if slr.ref_lib() is None:
        raise SystemExit("build oracle/_ref first: make -C oracle")
    gold = json.load(open(os.path.join(HERE, "golden.json")))
    rng = np.random.default_rng(20260)
    srcs = {}
    for name in ("ap_lowpass20", "ap_c13_58", "ap_c13_64", "qp_modelA48", "lin_cplx31", "lin_real64", "qphs21"):
        h = gold[name]["h"]
        srcs[name] = np.array(h["re"]) + 1j * np.array(h["im"])
    k = np.arange(127) - 63
    srcs["sinc127"] = np.sinc(k / 8.0) * np.hamming(127)
    srcs["rand200"] = (rng.standard_normal(200) + 1j * rng.standard_normal(200)) * np.hanning(200)
    srcs["rand511"] = (rng.standard_normal(511) + 1j * rng.standard_normal(511)) * np.hanning(511)
    srcs["one_tap"] = np.array([1.0 + 0.5j])
    out = {}
    for name, h in srcs.items():
        if name == "rand511":
            levels = ((0.7071, 1e-9),)
        else:
            levels = ((0.26, 1e-13), (0.7071, 1e-9), (0.966, 2e-5))
        for peak, tol in levels:
            b = scaled(h, peak)
            a_c = slr.ref_b2a(b)
            a_in = np.conj(a_c)
            rf_c = slr.ref_cabc2rf(a_in, b)
            out["%s_p%03d" % (name, round(peak * 100))] = dict(
                n=len(b), peak=peak, tol_a=tol, b=cplx(b), a_c=cplx(a_c), rf_c=cplx(rf_c))
    with open(os.path.join(HERE, "slr_golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote %d cases" % len(out))
    mat = "/root/reference/rf_tools/mex5/new.mat"
    if os.path.exists(mat):
        import scipy.io as sio
        m = sio.loadmat(mat)
        nm = dict(source="rf_tools/mex5/new.mat of the reference (variables h, hn, rf, rfm): a 64-tap beta polynomial "
                         "and the RF its MEX chain produced")
        for k in ("h", "hn", "rf", "rfm"):
            nm[k] = cplx(m[k].ravel())
        with open(os.path.join(HERE, "slr_newmat.json"), "w") as f:
            json.dump(nm, f)


if __name__ == "__main__":
    main()
