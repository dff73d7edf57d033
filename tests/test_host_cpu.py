"""CPU tests of the product's host side (no GPU): the C-ABI library loads and exports every
symbol include/mbfir.h declares, the C++ problem assembly equals the oracle's row by row, and
argument errors / early failures follow the reference.  No compute entry point is called."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
from conftest import CASES, ROOT, WHICH, c13

import mbfir
from oracle import assemble


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mbfir.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mbfir_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = mbfir.load_library()
    names = _header_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), "libmbfir.so lacks %s" % name
        assert name in mbfir.SYMBOLS, "python binding lacks %s" % name
    assert sorted(mbfir.SYMBOLS) == names
    assert b"gfx950" in lib.mbfir_version()


def test_struct_layouts_match_the_header():
    # compare the ctypes mirrors with what a C compiler makes of include/mbfir.h
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "mbfir.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu", sizeof(mbfir_opts), sizeof(mbfir_info), offsetof(mbfir_opts, refine), offsetof(mbfir_info, gram_flop), sizeof(mbfir_job), offsetof(mbfir_job, info));return 0;}'
    exe = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mbfir_sizeof_%d" % os.getpid())
    r = subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=prog, text=True, capture_output=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True).stdout.split()
    os.remove(exe)
    assert [int(v) for v in out] == [ctypes.sizeof(mbfir.Opts), ctypes.sizeof(mbfir.Info), mbfir.Opts.refine.offset,
                                     mbfir.Info.gram_flop.offset, ctypes.sizeof(mbfir.Job), mbfir.Job.info.offset]
    o = mbfir.make_opts(grid_m=123, verbose=1)
    assert (o.grid_m, o.refine, o.verbose, o.max_iter) == (123, -1, 1, 0)


def _params(fn, args):
    if fn == "fir_ap_cvx":
        return args[4:6]
    if fn == "fir_qp_cvx":
        obj = np.atleast_1d(args[5]).astype(float)
        return [args[4]] + list(obj) + [0.0] * (2 - len(obj)) + [len(obj)]
    return [0.0]


@pytest.mark.parametrize("name", sorted(CASES))
def test_product_assembly_equals_oracle(name):
    fn, args = CASES[name]
    O = getattr(assemble, "assemble_" + fn)(*args)
    rc, P = mbfir.assemble_dense(WHICH[fn], args[0], args[1], args[2], args[3], _params(fn, args))
    assert rc == 0, P
    for k in ("l", "nq3", "big"):
        assert O[k] == P[k]
    assert O["G"].shape == P["G"].shape
    scale = np.abs(O["G"]).max()
    assert np.abs(O["G"] - P["G"]).max() <= 4e-15 * scale
    assert np.abs(O["h"] - P["h"]).max() <= 4e-15 * max(1.0, np.abs(O["h"]).max())
    assert np.abs(O["c"] - P["c"]).max() <= 1e-12 * max(1.0, np.abs(O["c"]).max())


@pytest.mark.parametrize("n,grid_m", [(16, 100), (17, 333), (40, 1024)])
def test_grid_override_matches_oracle(n, grid_m):
    f, a, d = c13(n)
    O = assemble.assemble_fir_ap_cvx(n, f, a, d, 0.1, 1e-3, grid_m=grid_m)
    rc, P = mbfir.assemble_dense(0, n, f, a, d, (0.1, 1e-3), grid_m=grid_m)
    assert rc == 0 and P["Mf"] == grid_m + 10
    assert np.abs(O["G"] - P["G"]).max() <= 1e-14 and np.abs(O["h"] - P["h"]).max() <= 1e-15
    fl, al, dl = [-1, -0.5, 0.3, 0.8], [0, 0, 1, 1], [0.1, 0.1]
    for nn in (9, 10):
        O = assemble.assemble_fir_linprog(nn, fl, al, dl, grid_m=grid_m)
        rc, P = mbfir.assemble_dense(2, nn, fl, al, dl, grid_m=grid_m)
        assert rc == 0 and np.abs(O["G"] - P["G"]).max() <= 1e-14


def test_duplicate_grid_points_and_zero_width_band():
    # a band edge on a linspace sample (w = -pi) duplicates that sample; a zero-width band takes a(1)
    f, a, d = [-1.0, -0.5, 0.25, 0.25, 0.6, 1.0], [0, 0, 0.8, 0.3, 0, 0], [0.05, 0.1, 0.05]
    O = assemble.assemble_fir_ap_cvx(12, f, a, d, 0.5, 1e-1)
    rc, P = mbfir.assemble_dense(0, 12, f, a, d, (0.5, 1e-1))
    assert rc == 0 and O["G"].shape == P["G"].shape
    assert np.abs(O["G"] - P["G"]).max() <= 1e-14 and np.abs(O["h"] - P["h"]).max() <= 1e-15


def test_error_codes_mirror_the_reference():
    spec = ([0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1])
    rc, msg = mbfir.assemble_dense(0, 10, *spec, params=(-1.0, 1e-3))
    assert rc == mbfir.E_ARG and "invalid input of obj" in msg                   # fir_ap_cvx.m:171-173
    rc, msg = mbfir.assemble_dense(1, 10, *spec, params=(1.0, 1.0, 2.0, 3))
    assert rc == mbfir.E_ARG and "invalid input of obj" in msg                   # fir_qp_cvx.m:194-196
    rc, msg = mbfir.assemble_dense(2, 32, [0, 0.2, 0.3, 1], [0, 0, 1, 1], [0.01, 0.01])
    assert rc == mbfir.EARLY_FAIL and "fs/2" in msg                              # ss/fir_linprog.m:66-75
    rc, msg = mbfir.assemble_dense(3, 11, [0, 0.2, 0.4, 1], [1, 0.5, 0, 0], [0.1, 0.1])
    assert rc == mbfir.E_ARG and "sloped" in msg                                 # ss/fir_qprog_phs.m:53-57
    rc, msg = mbfir.assemble_dense(3, 11, [0, 0.2, 0.4, 1], [0.05, 0.05, 0, 0], [0.1 * np.exp(0.2j), 0.1])
    assert rc == mbfir.E_ARG and "straddling" in msg                             # ss/fir_qprog_phs.m:74-80
    rc, msg = mbfir.assemble_dense(3, 22, [-1, -0.6, -0.2, 0.2], [1, 1, 0, 0], [0.05 * np.exp(0.3j), 0.02])
    assert rc == mbfir.EARLY_FAIL                                                # ss/fir_qprog_phs.m:193-202


def test_python_mirror_argument_checks():
    with pytest.raises(ValueError, match="not enough input"):
        mbfir.fir_ap_cvx(10, None, None, None)
    with pytest.raises(ValueError, match="inconsistent"):
        mbfir.fir_ap_cvx(10, [0, 0.2, 0.4], [1, 1, 0], [0.1], ctx=object())
    with pytest.raises(ValueError, match="invalid input of obj"):
        mbfir.fir_qp_cvx(10, [0, 0.2, 0.4, 1], [1, 1, 0, 0], [0.1, 0.1], 1.0, [1, 2, 3], ctx=object())


def test_no_silent_cpu_fallback():
    """Without a GPU the product must refuse to run -- there is no CPU path behind the API."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(mbfir.MbfirError, match="no HIP device"):
        mbfir.Context(0)
    src = open(os.path.join(ROOT, "multiband-rf-pulse-design_amd", "__init__.py")).read()
    assert "oracle" not in src.replace("the oracle", "").replace("oracle's", "")   # product never imports the checker
    for fname in os.listdir(os.path.join(ROOT, "multiband-rf-pulse-design_amd", "csrc")):
        if fname.endswith((".cpp", ".hip", ".h")):
            body = open(os.path.join(ROOT, "multiband-rf-pulse-design_amd", "csrc", fname)).read()
            # comments may cite the oracle file a routine mirrors; nothing may include, open or load anything under oracle/
            for cited in ("oracle/conic_ipm.py", "oracle/assemble.py", "oracle/ddlin.c", "oracle/designers.py"):
                body = body.replace(cited, "")
            assert "#include \"../../oracle" not in body and "oracle/" not in body


def test_mex_gateway_compiles_against_a_stub_header():
    """matlab/mbfir_mex.c cannot be built for real (no MATLAB); check it is valid C against a
    minimal stand-in for mex.h that declares only the API calls the gateway uses."""
    stub = os.path.join(ROOT, "tests", "stubs")
    for name in ("mbfir_mex.c", "mbfir_slr_mex.c", "mbfir_bloch_mex.c"):
        src = os.path.join(ROOT, "matlab", name)
        for flags in ([], ["-DMX_HAS_INTERLEAVED_COMPLEX=1"]):
            r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-I", stub, "-I", os.path.join(ROOT, "include")] + flags + [src],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr


def test_h1qp_fixture_is_a_certified_optimum():
    """tests/golden/h1qp_golden.json (BASELINE config 3 read literally, made by make_golden_h1qp.py): the stored oracle
    results carry their own primal-dual certificates, re-evaluated in plain NumPy when the fixture was generated."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "h1qp_golden.json")) as fh:
        g = json.load(fh)
    assert set(g) == {"h1qp_384_6144", "h1qp_512_16384"}
    for rec in g.values():
        c = rec["certificate"]
        assert rec["status"] == 0 and rec["chol_fixes"] == 0
        assert c["pres"] <= 1e-8 and c["dres"] <= 1e-8 and c["s_outside"] <= 1e-12 and c["z_outside"] <= 1e-7
        assert abs(c["pcost"] - c["dcost"]) <= 1e-8 * abs(c["pcost"]) and abs(c["sz"]) <= 1e-8 * abs(c["pcost"])
        assert len(rec["h_re"]) == rec["n"] and abs(rec["pcost"] - c["pcost"]) <= 1e-12 * abs(c["pcost"])


def test_refine_is_clamped_and_new_options_exist():
    o = mbfir.make_opts(refine=20, ddkkt=1, lanes=8)
    assert o.refine == 20 and o.ddkkt == 1 and o.lanes == 8           # the clamp to 8 sweeps happens inside the library
    with pytest.raises(TypeError):
        mbfir.make_opts(no_such_option=1)


def test_bench_untimed_legs_run_in_a_child_whose_failure_cannot_cost_the_metric_line():
    """bench.py's untimed legs (BASELINE configs 3 and 4, the heterogeneous batch) run in a fresh child process; whatever
    happens to it -- here: there is no GPU, so the child's first context raises -- comes back as an "error" entry, never as an
    exception or a lost line of the parent."""
    sys.path.insert(0, ROOT)
    import bench
    res = bench.other_configs_in_child(0, 2, timeout_s=120)
    assert isinstance(res, dict) and "error" in res and ("child exit code" in res["error"] or "GPU" in res["error"] or "Error" in res["error"])



def test_generated_assembly_has_no_select_on_an_undefined_scalar_condition(tmp_path):
    """Round 6 met a backend miscompile (ROCm 7.2, gfx950): a uniform select on a vector compare came out as `v_cmp_* vcc ... ;
    s_cselect_b32 ...` with nothing defining SCC in between, so a flag the device solver branches its update on held a stale condition
    (solver.hip k_scal_step; the symptom: a corrected step taken along the uncorrected direction).  tools/scan_scc.py recognises that
    shape; every device source of the package is compiled to assembly here (hipcc cross-compiles without a GPU) and must be free of it."""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "multiband-rf-pulse-design_amd", "csrc")
    outs = []
    procs = []
    for name in sorted(f for f in os.listdir(csrc) if f.endswith(".hip")):
        out = str(tmp_path / (name[:-4] + ".s"))
        outs.append(out)
        procs.append(subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(csrc, name), "-o", out],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, err.decode()[-800:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_scc.py")] + outs, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("0 suspicious sites"), r.stdout[-1500:]
