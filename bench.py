#!/usr/bin/env python3
"""Headline benchmark: FIR designs/sec (and IPM iterations/sec) of the arbitrary-phase SOCP designer
at n=512 taps, m=16384 grid points (BASELINE.json metric), on N GPUs of one node.

A "step" is one batch of `--designs` (default 32) DISTINCT designs through the C ABI (mbfir_solve_batch:
host assembly, every IPM iteration on the GPU, spectral factorisation): the S-C13 bSSFP spec
(bSSFP_pulse_sb_mb.m:9-52) in the fixed-duration regime, obj=0.1, swept over 16 end-spike bounds Peak
(bSSFP_pulse_diff_Peak.m:68 sweeps Peak) x ripple pairs -- the shape of the reference's outer
loops.  Designs of one shape advance in LOCK STEP (`--lanes` per unit: one stream, one launch per phase, the
design index a grid dimension), units are spread over `--streams` contexts.  Inputs are a few dozen doubles,
so "inputs resident in HBM" is trivially true; the timed region includes the PCIe hand-over of the specs and
of the taps.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank designs its own batches
(independent designs shard with no data-path collective, SURVEY 8e "replicas only"): scaling weak, value =
N * designs * K / max-over-ranks time.  With N > 1 the line also carries "shard": BASELINE config 5 (n=2048,
m=131072) with the frequency rows of ONE design split over the ranks and the solver's own RCCL all-reduces
per iteration (strong scaling) -- `--mode shard` makes that the primary metric instead.

One JSON line on stdout (rank 0).
"""
import argparse
import gc
import glob
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MATRIX_TF = 78.6     # AMD's public MI355X fp64 matrix figure; the local hardware guide lists none


def sweep_jobs(mbfir, n, count):
    """`count` distinct designs: 16 Peak values log-spaced in [1e-4, 1e-2] x ripple pairs (0.01, 0.005) 2^(j/4)."""
    import numpy as np
    peaks = np.logspace(-4, -2, 16)
    jobs = []
    for q in range(count):
        j, p = divmod(q, 16)
        f, a, d = mbfir.spec.spec_c13_bssfp(n, d1=0.01 * 2 ** ((j % 16) / 4), d2=0.005 * 2 ** ((j % 16) / 4))
        jobs.append(("fir_ap_cvx", (n, f, a, d, 0.1, float(peaks[p]))))
    return jobs


def _cpu_quota():
    """CPUs the cgroup lets this process use at once (cpu.max quota / period), None when unlimited or unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        return None


def cpu_baseline(job, grid_m, gpu_iters, iters_cpu):
    """The oracle's C++ / OpenMP solver (oracle/cpu_ipm.cpp: the same algorithm as the GPU solver on the DENSE normal
    matrix, register-blocked Gram product on the widest FMA the host has: rebuilt with -march=native on the box it is
    timed on) on this box's host cores, on ONE design of the batch.
    iters_cpu < 0: the whole design to convergence on all cores, plus a 1-thread sample of 2 iterations extrapolated
    to the same iteration count; > 0: assembly + that many IPM iterations on all cores, extrapolated (labelled)."""
    from oracle import assemble, cpu_ipm
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    isa = cpu_ipm.use_native_build() or (cpu_ipm.isa() + " -- portable x86-64-v3 build, the -march=native rebuild failed on this box")
    n, f, a, d, obj, peak = job[1]
    t0 = time.perf_counter()
    P = assemble.assemble_fir_ap_cvx(n, f, a, d, obj, peak, grid_m)
    t_asm = time.perf_counter() - t0
    full = iters_cpu < 0
    run = lambda **kw: cpu_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], **kw)
    # thread count: a box may show more CPUs than its share lets run at once (256 visible, 16 granted on the GPU pool), so
    # the count is picked by timing one iteration at each candidate -- the baseline is the CPU at its best
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    trials = {}
    for th in sorted({t for t in (4, 8, 16, 32, 64, 128, avail) if t <= avail}):
        trials[th] = run(max_iter=1, threads=th)["seconds_total"]
        if len(trials) >= 2 and trials[th] > 1.5 * min(trials.values()):
            break
    best_threads = min(trials, key=trials.get)
    r = run(threads=best_threads, **({} if full else {"max_iter": iters_cpu}))
    cores = r["threads"]
    if full:
        t_design = t_asm + r["seconds_total"]
        t_iter = r["seconds_total"] / (r["iters"] + 1)
        sample = ("oracle/cpu_ipm.cpp (C++ OpenMP dense conic IPM, same algorithm) run to convergence on the first design of the "
                  "batch on %d threads: %d iterations, status %d, %.1f s solve + %.1f s NumPy assembly"
                  % (cores, r["iters"], r["status"], r["seconds_total"], t_asm))
        one = run(max_iter=2, threads=1)
        t_iter1 = one["seconds_total"] / 3                # the initial point costs one factorisation + solve
        single = {"cores": 1, "s_per_iteration": t_iter1, "value": 1.0 / (t_asm + t_iter1 * (r["iters"] + 1)),
                  "sample": "2 IPM iterations on 1 thread, EXTRAPOLATED to the %d iterations of the all-core run" % r["iters"]}
    else:
        t_iter = r["seconds_total"] / (iters_cpu + 1)
        t_design = t_asm + t_iter * (gpu_iters + 1)
        single = None
        sample = ("oracle/cpu_ipm.cpp on %d threads: assembly %.1f s + first %d of %d IPM iterations (%.2f s each), EXTRAPOLATED "
                  "to the full design" % (cores, t_asm, iters_cpu, gpu_iters, t_iter))
    return {"value": 1.0 / t_design, "unit": "designs/s", "cores": int(cores), "kind": "port", "sample": sample,
            "extrapolated": not full, "s_per_iteration": t_iter, "iters_per_s": 1.0 / t_iter,
            "build": "g++ -O3 -march=native -fopenmp, compiled on this box: " + isa,
            "gram_fraction": r["seconds_factor"] / max(r["seconds_total"], 1e-9), "single_thread": single,
            "thread_trials_s_per_2_factorisations": {str(k): v for k, v in trials.items()}, "cpus_visible": avail, "cpu_quota": _cpu_quota(),
            "pcost": float(r["pcost"]) if full else None}


def s_lp_baseline(mbfir, ctx):
    """BASELINE config 1 (S-LP: n=64 linear-phase low-pass, fir_linprog form, 512 grid points): the GPU design beside
    scipy.optimize.linprog(method="highs") and the oracle's C++ solver on the identical program;
    None when SciPy is missing on the box."""
    try:
        from scipy.optimize import linprog
    except Exception:
        return None
    from oracle import assemble, cpu_ipm
    import numpy as np
    args = (64, [0, 0.2, 0.3, 1], [1, 1, 0, 0], [0.01, 0.01])
    opts = mbfir.make_opts(grid_m=512)
    mbfir.fir_linprog(*args, opts=opts, ctx=ctx)                     # warm (allocations of this shape)
    t0 = time.perf_counter()
    h, status, info = mbfir.fir_linprog(*args, opts=opts, ctx=ctx, info=True)
    t_gpu = time.perf_counter() - t0
    P = assemble.assemble_fir_linprog(*args, 512)
    t0 = time.perf_counter()
    rh = linprog(P["c"], A_ub=P["G"], b_ub=P["h"], bounds=[(None, None)] * len(P["c"]), method="highs")
    t_highs = time.perf_counter() - t0
    rc = cpu_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], threads=1)
    return {"workload": "S-LP: fir_linprog(64, [0 .2 .3 1], [1 1 0 0], [.01 .01]), grid_m=512: %d rows x %d unknowns" % P["G"].shape,
            "gpu_ms": t_gpu * 1e3, "gpu_status": status, "gpu_iters": info["iters"], "highs_ms": t_highs * 1e3,
            "cpp_1thread_ms": rc["seconds_total"] * 1e3, "objective_gpu_highs_cpp": [info["pcost"], float(rh.fun), rc["pcost"]],
            "note": "a 1028 x 32 LP is launch-latency bound on the GPU; this leg is the reference's CPU-runnable case, not a throughput claim"}


def other_configs(mbfir, ctxs):
    """BASELINE configs 3 and 4 as batches, AFTER the timed region (rank 0, one GPU): not the metric, but measured in the same run
    so that the record holds them.  Config 4: bSSFP_pulse_diff_Peak's sweep, 256 designs n=200 m=4096, lock-step units of 32 on
    the bench's streams.  Config 3: specsat_H1_dualband through fir_qp_cvx as written (k=120, obj=1e6, n=512, m=16384; the
    extended-precision KKT solve runs one design per stream), 8 designs whose ripples differ by 2 % steps, 8 streams."""
    out = {}
    jobs4 = sweep_jobs(mbfir, 200, 256)
    o4 = mbfir.make_opts(grid_m=4096, lanes=32)
    mbfir.solve_batch(jobs4, ctxs=ctxs, opts=o4)        # warm EVERY context (a fresh child process: arenas and pinned staging are allocated on first use; 64 jobs = two units of 32 reached two of the four)
    t = time.perf_counter()
    res = mbfir.solve_batch(jobs4, ctxs=ctxs, opts=o4, info=True)
    dt = time.perf_counter() - t
    out["config4_sweep_256_designs_n200_m4096"] = {"designs_per_s": 256 / dt, "solved": sum(1 for r in res if r[1] == "Solved"),
                                                   "lanes": res[0][2]["lanes"], "streams": len(ctxs),
                                                   "ipm_iters_per_design": sum(r[2]["iters"] for r in res) / 256.0}
    # heterogeneous batch: 64 S-RAND specs (k in 2..8 bands, 64 different edge sets) at the
    # headline size -- what the reference's callers produce (bisection probes, sweeps over specs), where the headline's 64
    # designs share their band edges.  Designs of one order share lock-step units whatever their grids (heterogeneous units).
    # Feasible draws only, established on the device (the designers' own verdict): the candidates are designed once -- which
    # also warms the allocations -- and the first 64 that solve are timed as one batch.
    try:
        cand = []
        for seed in range(160):
            try:
                fr, ar, dr = mbfir.spec.spec_rand(512, seed)
            except ValueError:
                continue
            cand.append(("fir_ap_cvx", (512, list(fr), list(ar), list(dr), 0.1, 1e-3)))
        oh = mbfir.make_opts(grid_m=16384, lanes=16)
        keep, tried, nfailed = [], 0, 0
        while len(keep) < 64 and tried < len(cand):
            chunk = cand[tried:tried + 64]
            tried += len(chunk)
            ok = [r[1] == "Solved" for r in mbfir.solve_batch(chunk, ctxs=ctxs, opts=oh)]
            keep += [job for job, good in zip(chunk, ok) if good]
            nfailed += sum(1 for good in ok if not good)
        keep = keep[:64]
        if len(keep) == 64:
            t = time.perf_counter()
            res = mbfir.solve_batch(keep, ctxs=ctxs, opts=oh, info=True)
            dt = time.perf_counter() - t
            out["heterogeneous_64"] = {
                "designs_per_s": 64 / dt, "ms_per_batch": dt * 1e3, "solved": sum(1 for r in res if r[1] == "Solved"),
                "ipm_iters_per_design": sum(r[2]["iters"] for r in res) / 64.0, "lanes_per_unit": sorted({r[2]["lanes"] for r in res}),
                "distinct_shapes_rows_freq": len({(r[2]["n_rows"], r[2]["n_freq"]) for r in res}), "bands": sorted({len(j[1][3]) for j in keep}),
                "draws_tried": tried, "draws_failed": nfailed, "streams": len(ctxs),
                "workload": "64 S-RAND specs (spec_rand seeds in order, feasible ones), fir_ap_cvx(n=512, obj=0.1, Peak=1e-3), grid_m=16384"}
        else:
            out["heterogeneous_64"] = {"error": "only %d of %d S-RAND draws solved" % (len(keep), tried)}
    except Exception as e:                                  # noqa: BLE001
        out["heterogeneous_64"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # the DENSE path in lock-step units: north_star's own formulation -- materialised trig matrix, A'DA on the
    # fp64 matrix cores per build -- as throughput, beside the dense CPU baseline's throughput (main() forms the ratio)
    try:
        jd = sweep_jobs(mbfir, 512, 16)
        od = mbfir.make_opts(grid_m=16384, dense_trig=1, lanes=8)
        mbfir.solve_batch(jd, ctxs=ctxs, opts=od)
        t = time.perf_counter()
        res = mbfir.solve_batch(jd, ctxs=ctxs, opts=od, info=True)
        dt = time.perf_counter() - t
        out["dense_batch16"] = {"designs_per_s": 16 / dt, "ms_per_batch": dt * 1e3, "solved": sum(1 for r in res if r[1] == "Solved"),
                                "lanes_per_unit": sorted({r[2]["lanes"] for r in res}), "streams": len(ctxs), "lattice": sorted({r[2]["lattice"] for r in res}),
                                "ipm_iters_per_design": sum(r[2]["iters"] for r in res) / 16.0,
                                "workload": "the first 16 designs of the headline sweep with opts.dense_trig=1 (k_build_A1, k_gram per build, k_amulti / k_atmulti per product)"}
    except Exception as e:                                  # noqa: BLE001
        out["dense_batch16"] = {"error": "%s: %s" % (type(e).__name__, e)}
    f, a, d = mbfir.spec.spec_h1_dualband(512)
    jobs16 = [("fir_qp_cvx", (512, f, a, [x * (1.0 + 0.02 * q) for x in d], 120.0, 1e6)) for q in range(16)]
    jobs3 = jobs16[:8]
    o3 = mbfir.make_opts(grid_m=16384)
    pool = list(ctxs) + [mbfir.Context(ctxs[0].device) for _ in range(max(0, 8 - len(ctxs)))]
    try:
        # round 5: the extended-precision solve takes lock-step units -- 16 designs over the bench's own streams (the library forms
        # units of 16 / streams lanes); the batch of 8 on 8 streams (one design per stream: rounds 2-4's figure) stays beside it
        mbfir.solve_batch(jobs16, ctxs=ctxs, opts=o3)
        t = time.perf_counter()
        res16 = mbfir.solve_batch(jobs16, ctxs=ctxs, opts=o3, info=True)
        dt16 = time.perf_counter() - t
        mbfir.solve_batch(jobs3[:len(pool)], ctxs=pool, opts=o3)
        t = time.perf_counter()
        res = mbfir.solve_batch(jobs3, ctxs=pool, opts=o3, info=True)
        dt = time.perf_counter() - t
        t = time.perf_counter()
        _, st1, i1 = mbfir.fir_qp_cvx(*jobs3[0][1], opts=o3, ctx=pool[0], info=True)
        dt1 = time.perf_counter() - t
        out["config3_fir_qp_cvx_h1_dualband_n512_m16384"] = {
            "designs_per_s_batch16": 16 / dt16, "batch16_streams": len(ctxs), "batch16_lanes_per_unit": sorted({r[2]["lanes"] for r in res16}),
            "batch16_solved": sum(1 for r in res16 if r[1] == "Solved"),
            "designs_per_s": 8 / dt, "batch": 8, "streams": len(pool), "solved": sum(1 for r in res if r[1] == "Solved"),
            "ipm_iters": [r[2]["iters"] for r in res], "extended_precision_iters": [r[2]["dd_iters"] for r in res],
            "one_design_alone_s": dt1, "one_design_status": st1,
            "extended_precision_form": {0: "capacitance (saddle-point) form, plain double on the matrix cores (capkkt.hip)", 1: "double-double factorisation (ddlin.hip)"}.get(i1.get("dd_form"), "none"),
            # the three fp64 MFMA products of the capacitance form (Yt = U M', Zt = Yt M, S = Yt Yt' + X^-1), HIP events around them
            "capacitance_products": None if not i1.get("ms_cap") else {
                "kernel": "k_cap_gemm<0|1|2> (64 x 64 tiles, v_mfma_f64_16x16x4_f64; triangular M: lower tiles only)", "bound": "mfma",
                "achieved": i1["cap_flop"] / (i1["ms_cap"] * 1e-3) / 1e12, "peak": PEAK_FP64_MATRIX_TF, "unit": "TFLOP/s",
                "frac": i1["cap_flop"] / (i1["ms_cap"] * 1e-3) / 1e12 / PEAK_FP64_MATRIX_TF, "traffic": None,
                "builds": i1["dd_iters"], "ms_per_build": i1["ms_cap"] / max(1, i1["dd_iters"]), "flop_per_build": i1["cap_flop"] / max(1, i1["dd_iters"]),
                "k_max": i1["dd_kmax"]}}
    finally:
        for c in pool[len(ctxs):]:
            c.close()
    return out


def other_configs_in_child(device, nstream, timeout_s=900):
    """The untimed legs in a FRESH child process: a native fault there (they include the 8-context extended-precision
    batch) cannot be caught by try/except and would take the already-measured metric line with it.  The child is started from this
    process (subprocess: fork + exec in the child, never an exec of the GPU-initialised parent) and prints one JSON object; a
    non-zero exit, a timeout or unparsable output becomes an "error" entry."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--other-configs-child", "--device", str(device), "--streams", str(nstream)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "the child running the untimed legs did not finish within %d s" % timeout_s}
    except Exception as e:                                  # noqa: BLE001
        return {"error": "could not start the child for the untimed legs: %s: %s" % (type(e).__name__, e)}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "child exit code %d" % r.returncode, "stderr_tail": r.stderr[-600:]}
    try:
        return json.loads(lines[-1])
    except Exception as e:                                  # noqa: BLE001
        return {"error": "unparsable child output: %s" % e, "stdout_tail": r.stdout[-300:]}


def other_configs_child(device, nstream):
    """entry of the child process: its own contexts, the legs of other_configs, one JSON object on stdout"""
    import mbfir
    ctxs = [mbfir.Context(device) for _ in range(nstream)]
    try:
        try:
            res = other_configs(mbfir, ctxs)
        except Exception as e:                              # noqa: BLE001
            res = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(res), flush=True)
    finally:
        for c in ctxs:
            c.close()


def pmc_traffic():
    """HBM-side bytes per k_chol_dag launch (one launch = one factorisation of a lock-step unit) from the newest committed PMC
    passes (tools/rocprof_summary.py writes profiles/rNN_pmc_traffic.json with the commit it was measured at); None when no such
    file exists."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        d = json.load(fh)
    d["file"] = "profiles/" + os.path.basename(files[-1])
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--taps", dest="n", type=int, default=512)
    ap.add_argument("--grid-m", type=int, default=16384)
    ap.add_argument("--mode", choices=["batch", "shard"], default="batch")
    ap.add_argument("--designs", type=int, default=64, help="distinct designs per step and rank")
    ap.add_argument("--lanes", type=int, default=16, help="designs per lock-step unit (mbfir_opts.lanes; 1 = one design per stream)")
    ap.add_argument("--streams", type=int, default=4, help="contexts / HIP streams the units are spread over")
    ap.add_argument("--cpu-iters", type=int, default=-1, help="cpu_baseline leg: -1 the oracle to convergence on one design "
                    "(1-3 minutes), k > 0 the first k iterations extrapolated, 0 skip")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the untimed legs for BASELINE configs 3 and 4")
    ap.add_argument("--shard-n", type=int, default=2048)
    ap.add_argument("--shard-grid-m", type=int, default=131072)
    ap.add_argument("--no-shard", action="store_true", help="N > 1: skip the row-sharded config-5 leg")
    ap.add_argument("--no-shard-dense", action="store_true", help="N > 1: skip the dense-path half of the row-sharded leg")
    ap.add_argument("--shard-timeout", type=int, default=360, help="N > 1: seconds the row-sharded leg may take before the line "
                    "is printed without it")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a box with fewer GPUs than ranks: ranks then share devices and the solver's "
                    "reductions go through the host hook instead of its own RCCL communicator)")
    ap.add_argument("--dense", action="store_true", help="materialised trig matrix + dense MFMA Gram (opts.dense_trig) "
                    "instead of the default lattice (matrix-free) mode; one design per stream")
    ap.add_argument("--other-configs-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--device", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.other_configs_child:
        other_configs_child(args.device, max(1, args.streams))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves -- a child `python -m torch.distributed.run`, spawned
        # BEFORE anything in this process touches the GPU (no exec from a process that has initialised it); the ranks' output
        # (rank 0's JSON line) passes straight through, the child's exit code is ours
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend)
    if args.gpus != world:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE %d; launch with torch.distributed.run\n" % (args.gpus, world))
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    import mbfir
    nstream = max(1, args.streams)
    ctxs = [mbfir.Context(local_rank) for _ in range(nstream)]
    ctx = ctxs[0]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def wire_shard(c):
        """reductions of a row-sharded solve: the solver's own RCCL communicator, or (gloo rehearsal) a host hook"""
        if args.backend == "nccl":
            c.init_comm()
        else:
            def hook(ptr, count, op, _t=torch):
                import torch.distributed as dist
                t = mbfir.device_tensor(ptr, count)
                cc = t.cpu()
                dist.all_reduce(cc, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
                t.copy_(cc)
                _t.cuda.synchronize()
                return 0
            c.set_allreduce(hook)

    def shard_leg(n, grid_m, steps, warmup, dense=None):
        """ONE design, its frequency rows split over the ranks (strong scaling).  dense: the materialised trig matrix and the MFMA
        Gram product (north_star's own design: the packed lower triangle of the normal matrix is all-reduced per iteration)
        instead of the lattice path (the trigonometric moments are)."""
        dense = int(args.dense if dense is None else dense)
        f, a, d = mbfir.spec.spec_c13_bssfp(n)
        o = mbfir.make_opts(grid_m=grid_m, shard_rank=rank, shard_size=world, dense_trig=dense)
        info = None
        for _ in range(warmup):
            mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o, ctx=ctx, info=True)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            h, status, info = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o, ctx=ctx, info=True)
            if status != "Solved":
                raise RuntimeError("sharded design did not solve: %r" % (info,))
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        # the same design unsharded on this rank IN THE SAME MODE (no collective): what one GPU does alone, and the split of
        # the sharded solve into the part that shards (row work: Gram product / moments, G v, G'v, step lengths) and the part
        # that does not (the factorisation of the replicated N x N normal matrix, which every rank runs for itself)
        o1 = mbfir.make_opts(grid_m=grid_m, dense_trig=dense)
        mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o1, ctx=ctx, info=True)
        fence()
        t1 = time.perf_counter()
        _, st1, info1 = mbfir.fir_ap_cvx(n, f, a, d, 0.1, 1e-3, opts=o1, ctx=ctx, info=True)
        torch.cuda.synchronize()
        one_gpu_ms = (time.perf_counter() - t1) * 1e3
        fence()
        sharded_ms = el / steps * 1e3
        chol_ms = info1["ms_chol"]                          # the factorisations of the unsharded solve = the replicated part
        amdahl = one_gpu_ms / (chol_ms + (one_gpu_ms - chol_ms) / world) if one_gpu_ms > 0 else None
        return {"metric": "FIR designs/sec, n=%d taps m=%d, frequency rows of one design sharded x%d (%s path)" % (n, grid_m, world, "dense" if dense else "lattice"),
                "value": steps / el, "unit": "designs/s", "ms_per_design": sharded_ms, "scaling": "strong", "mode": "dense" if dense else "lattice",
                "iters": info["iters"], "collectives_per_iteration": info["collectives"] / max(1, info["iters"]),
                "collective_MB_per_iteration": info["collective_bytes"] / max(1, info["iters"]) / 1e6,
                "one_gpu_unsharded_ms": one_gpu_ms, "speedup_vs_one_gpu": one_gpu_ms / sharded_ms if st1 == "Solved" else None,
                "phase_split_ms": {"replicated_factorisation": chol_ms, "shardable_row_work_one_gpu": one_gpu_ms - chol_ms,
                                   "sharded_solve_total": sharded_ms},
                "amdahl_bound_speedup_at_this_n_gpus": amdahl,
                "note": "strong scaling of ONE design is bounded by the replicated factorisation (every rank factorises the same N x N "
                        "normal matrix itself -- lattice: from the all-reduced moments, dense: from the all-reduced packed lower "
                        "triangle); the weak-scaling batch figure above (independent designs, no collective) is the mode that approaches N x",
                "reductions": "ncclAllReduce on the solver stream (mbfir_comm_init)" if args.backend == "nccl" else "host hook (gloo rehearsal)"}

    shard_primary = args.mode == "shard" and world > 1
    if shard_primary:
        wire_shard(ctx)
        res = shard_leg(args.n, args.grid_m, args.steps, args.warmup)
        if rank == 0:
            out = {"metric": res["metric"], "value": res["value"], "unit": "designs/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": res["ms_per_design"], "higher_is_better": True, "scaling": "strong",
                   "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": "S-C13 bSSFP spec, fir_ap_cvx(n=%d, obj=0.1, Peak=1e-3), grid_m=%d, rows sharded x%d"
                                          % (args.n, args.grid_m, world)}, "shard": res, "roofline": None, "cpu_baseline": None}
            print(json.dumps(out), flush=True)
        for c in ctxs:
            c.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    lanes = max(1, args.lanes)                   # (the dense path batches too; the solver caps the lanes by the memory of the trig matrices)
    opts = mbfir.make_opts(grid_m=args.grid_m, dense_trig=int(args.dense), lanes=lanes)
    jobs = sweep_jobs(mbfir, args.n, args.designs)

    def step():
        res = mbfir.solve_batch(jobs, opts=opts, ctxs=ctxs, info=True)
        for h, status, info in res:
            if status != "Solved":
                raise RuntimeError("benchmark design did not solve: %r" % (info,))
        return [info for _, _, info in res]

    for _ in range(args.warmup):
        step()
    # the interpreter's cyclic collector: with torch imported a full collection walks ~1e6 objects (50-80 ms, seen as
    # single steps of 320 ms among 250 ms ones); the objects alive now go to the permanent generation, collections of
    # what the steps allocate stay on
    gc.collect()
    gc.freeze()
    fence()
    t0 = time.perf_counter()
    infos = [i for _ in range(args.steps) for i in step()]
    fence()
    elapsed = time.perf_counter() - t0
    ndesign = args.steps * args.designs          # designs this rank completed in the timed region
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- outside the timed region -----------------------------------------------------------------------
    # (1) one lock-step unit alone on the GPU: the HIP events around its k_chol_step / moment launches then see
    #     nothing else, and the figures agree with rocprofv3's kernel durations (profiles/)
    unit = mbfir.solve_batch(jobs[:lanes], opts=opts, ctxs=ctxs[:1], info=True)
    uinfo = unit[0][2]
    # (2) single-design latency, one stream
    t1 = time.perf_counter()
    _, _, solo = mbfir.fir_ap_cvx(*jobs[0][1], opts=mbfir.make_opts(grid_m=args.grid_m, dense_trig=int(args.dense)), ctx=ctx, info=True)
    torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) * 1e3
    # (3) the north-star kernel A'DA on the matrix cores is the dense path's (opts.dense_trig); one dense design
    #     keeps its live MFMA rate in the record and is the like-for-like partner of the (dense) CPU baseline
    dense_info = None
    if not args.dense:
        _, st_d, dense_info = mbfir.fir_ap_cvx(*jobs[0][1], ctx=ctx, info=True, opts=mbfir.make_opts(grid_m=args.grid_m, dense_trig=1))
        if st_d != "Solved":
            dense_info = None
    out = None
    if rank == 0:
        iters = sum(i["iters"] for i in infos)
        lattice = bool(infos[0]["lattice"])
        try:
            peak_mfma, peak_valu = mbfir.mfma_peak(ctx)
        except Exception:
            peak_mfma = peak_valu = float("nan")
        ul = max(1, uinfo["lanes"])
        # k_chol_dag: ONE launch per factorisation (Cholesky + triangular inverse) for all lanes of the unit at once.
        # Algorithmic flop per launch: lanes x 2/3 np^3 with the explicit inverse the solves use (an implementation
        # choice), lanes x 1/3 np^3 by SURVEY.md 8(d)'s F_chol = N^3 / 3 -- both fractions are printed.  Measured twice:
        # one unit alone on the GPU (HIP events around its launches; agrees with rocprofv3) and under the bench's own load
        # (the same events inside the timed batch, `streams` units in flight).
        chol_launches = uinfo["chol_launches"]
        chol_flop_per_launch = ul * uinfo["chol_flop"] * uinfo["builds"] / max(1, chol_launches)
        chol_avg_ms = uinfo["ms_chol"] / max(1, chol_launches)
        chol_ach = chol_flop_per_launch / (chol_avg_ms * 1e-3) / 1e12 if uinfo["ms_chol"] > 0 else 0.0
        # under load: every unit of the timed batch reports the summed device time of its factorisations
        seen, load_ms, load_launches, load_flop = set(), 0.0, 0, 0.0
        for i in infos:
            key = (i["ms_chol"], i["ms_solve"], i["lanes"])               # one entry per unit (its lanes share the figures)
            if key in seen or i["chol_launches"] <= 0:
                continue
            seen.add(key)
            load_ms += i["ms_chol"]; load_launches += i["chol_launches"]
            load_flop += max(1, i["lanes"]) * i["chol_flop"] * i["builds"]
        load_avg_ms = load_ms / max(1, load_launches)
        load_ach = load_flop / max(load_ms * 1e-3, 1e-12) / 1e12
        pmc = pmc_traffic()
        per_step = chol_launches > uinfo["builds"] + 1                     # (MBFIR_CHOL_SPLIT=0..2: one launch per panel step)
        roof_chol = {"kernel": ("k_chol_step (one launch per 64-wide panel step" if per_step else "k_chol_dag (blocked Cholesky + triangular inverse of all "
                                "%d lanes of a lock-step unit in ONE launch: ticket-ordered tasks, per-tile dependency counters" % ul) +
                               "; tile products on v_mfma_f64_16x16x4_f64)", "bound": "mfma",
                     "achieved": chol_ach, "peak": PEAK_FP64_MATRIX_TF, "unit": "TFLOP/s", "frac": chol_ach / PEAK_FP64_MATRIX_TF,
                     "frac_survey_8d_factor_only": 0.5 * chol_ach / PEAK_FP64_MATRIX_TF,
                     "achieved_under_load": load_ach, "frac_under_load": load_ach / PEAK_FP64_MATRIX_TF, "avg_launch_ms_under_load": load_avg_ms,
                     "under_load_note": "per-unit rate with %d units in flight (each unit has the chip to itself only part of the time); "
                                        "whole-chip rate = this x the units that overlap" % nstream,
                     "traffic": (pmc or {}).get("k_chol_bytes_per_launch") if infos[0]["n_unknowns"] == 1024 and ul == (pmc or {}).get("lanes") and not per_step else None,
                     "traffic_source": None if pmc is None else "%s (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes, "
                                       "commit %s, %s lanes)" % (pmc.get("file"), pmc.get("commit"), pmc.get("lanes")),
                     "flop_per_launch": chol_flop_per_launch, "flop_per_launch_factor_only": 0.5 * chol_flop_per_launch,
                     "launches": chol_launches, "avg_launch_ms": chol_avg_ms, "lanes": ul,
                     "note": "two dependency chains per design (1024 sequential pivots; 16 inverse rows, each waiting for the one above); "
                             "the lanes of a lock-step unit fill the CUs the chains of one design leave idle; measured on one unit alone on the GPU",
                     "peak_source": "AMD public MI355X fp64 matrix figure (not in MI355X_MICROARCH.md)",
                     "register_only_mfma_loop_tflops": peak_mfma, "register_only_fma_loop_tflops": peak_valu,
                     "microbenchmark_note": "a register-only loop of independent v_mfma_f64_16x16x4 (8 accumulators per wave, 4 waves per SIMD) "
                                            "runs power-capped at a lower clock than kernels that wait on memory; it is NOT a peak (k_gram sustains more)"}
        gram_flop = uinfo["gram_flop"]
        if lattice:
            gram_ach = ul * gram_flop * uinfo["builds"] / (uinfo["ms_gram"] * 1e-3) / 1e12 if uinfo["ms_gram"] > 0 else 0.0
            roof_gram = {"kernel": "k_trig_moments + fold + k_assemble_H_lat (normal matrix from trigonometric moments, fp64 VALU)",
                         "bound": "valu", "achieved": gram_ach, "peak": PEAK_FP64_MATRIX_TF, "unit": "TFLOP/s",
                         "frac": gram_ach / PEAK_FP64_MATRIX_TF, "traffic": None, "flop_per_build": ul * gram_flop, "builds": uinfo["builds"],
                         "avg_build_ms": uinfo["ms_gram"] / max(1, uinfo["builds"]), "lanes": ul,
                         "note": "replaces the dense Gram products (17.2 GFLOP per design and build on the matrix cores) by "
                                 "%.2f GFLOP of recurrences; peak = fp64 vector peak (same figure as the matrix peak)" % (gram_flop / 1e9)}
        else:
            # dense path: the events bracket the k_gram launches of ALL live lanes of a build (one launch per lane and weight
            # vector); lanes that finish early drop out, so with ul > 1 this is a lower bound of the per-launch rate -- the
            # exact single-launch figure is the dense single-design leg below (roofline_other)
            launches = uinfo["gram_launches"] * ul
            flop_per_launch = gram_flop / max(1, uinfo["gram_launches"] // max(1, uinfo["builds"]))
            gram_ach = flop_per_launch / (uinfo["ms_gram"] / max(1, launches) * 1e-3) / 1e12 if uinfo["ms_gram"] > 0 else 0.0
            roof_gram = {"kernel": "k_gram (A' D A, v_mfma_f64_16x16x4_f64)", "bound": "mfma", "achieved": gram_ach,
                         "peak": PEAK_FP64_MATRIX_TF, "unit": "TFLOP/s", "frac": gram_ach / PEAK_FP64_MATRIX_TF,
                         "traffic": None, "flop_per_launch": flop_per_launch, "launches": launches,
                         "avg_launch_ms": uinfo["ms_gram"] / max(1, launches),
                         "peak_source": "AMD public MI355X fp64 matrix figure (not in MI355X_MICROARCH.md)"}
        # `roofline` is the factorisation's entry whatever the size: it is the dominant kernel of the metric's workload (35 % of the
        # kernel time at n=512 against 13 % for the moment kernels) and its bound is one the contract knows ("hbm" | "mfma").  Until
        # round 3 the entry with the larger device time was picked, which at toy sizes (np = 256: four panel steps) flipped with the
        # box between this and the moment build -- fp64 VECTOR work, labelled "valu", outside the contract.  The moment build keeps
        # its own label in roofline_other; both carry their share of the unit's device time.
        for rf_, ms_ in ((roof_chol, uinfo["ms_chol"]), (roof_gram, uinfo["ms_gram"])):
            rf_["share_of_unit_solve_ms"] = ms_ / max(uinfo["ms_solve"], 1e-12)
        dominant, other = roof_chol, roof_gram
        others = [other]
        if dense_info is not None and dense_info["gram_launches"] > 0:
            fl = dense_info["gram_flop"] / max(1, dense_info["gram_launches"] // max(1, dense_info["builds"]))
            avg = dense_info["ms_gram"] / dense_info["gram_launches"]
            ach_d = fl / (avg * 1e-3) / 1e12
            others.append({"kernel": "k_gram (A' D A, v_mfma_f64_16x16x4_f64) -- dense path (opts.dense_trig=1), one design "
                                     "after the timed region", "bound": "mfma", "achieved": ach_d, "peak": PEAK_FP64_MATRIX_TF,
                           "unit": "TFLOP/s", "frac": ach_d / PEAK_FP64_MATRIX_TF, "traffic": (pmc or {}).get("k_gram_bytes_per_launch"),
                           "flop_per_launch": fl, "launches": dense_info["gram_launches"], "avg_launch_ms": avg,
                           "dense_design_ms": dense_info["ms_total"]})
        out = {
            "metric": "FIR designs/sec, n=%d taps m=%d arbitrary-phase SOCP (fir_ap_cvx form)" % (args.n, args.grid_m),
            "value": world * ndesign / elapsed, "unit": "designs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "S-C13 bSSFP 5-band spec, fixed-duration regime, fir_ap_cvx(n=%d, obj=0.1, Peak) swept over 16 Peak "
                                   "values x %d ripple pairs, grid_m=%d (+10 band edges); a step is one batch of %d DISTINCT designs per rank "
                                   "(mbfir_solve_batch: lock-step units of %d designs over %d HIP streams)"
                                   % (args.n, max(1, args.designs // 16), args.grid_m, args.designs, lanes, nstream),
                       "designs_per_step_per_rank": args.designs, "lanes": lanes, "streams": nstream,
                       "single_design_latency_ms": latency_ms,
                       "n_taps": args.n, "grid_m": args.grid_m, "unknowns": infos[0]["n_unknowns"], "rows": infos[0]["n_rows"],
                       "mode": args.mode, "trig": "lattice (matrix-free)" if lattice else "dense (materialised trig matrix, MFMA Gram)",
                       "parallelism": "independent designs x%d" % world},
            "ipm_iters_per_design": iters / ndesign,
            "ipm_iters_per_s": world * iters / elapsed,
            # passes over the frequency rows per iteration of a lock-step unit (mbfir_info.gv_passes / gtv_passes: G v and G'v launches,
            # a two-vector pass counts once) and the centrality corrector's share (round 6)
            "gv_passes_per_iter": sum(i["gv_passes"] / max(1, i["iters"]) for i in infos) / len(infos),
            "gtv_passes_per_iter": sum(i["gtv_passes"] / max(1, i["iters"]) for i in infos) / len(infos),
            "corrected_directions_taken_frac": sum(i["correctors_taken"] for i in infos) / max(1, sum(i["correctors"] for i in infos)),
            "ms_breakdown_single_stream": {"assemble": solo["ms_assemble"], "solve": solo["ms_solve"], "normal_matrix": solo["ms_gram"],
                                           "cholesky_inverse": solo["ms_chol"], "spectral_factor": solo["ms_post"], "iters": solo["iters"]},
            "ms_breakdown_lockstep_unit": {"lanes": ul, "solve": uinfo["ms_solve"], "normal_matrix": uinfo["ms_gram"],
                                           "cholesky_inverse": uinfo["ms_chol"], "builds": uinfo["builds"]},
            "roofline": dominant,
            "roofline_other": others,
        }
        if world == 1 and not args.dense and not args.no_other_configs:
            out["other_baseline_configs"] = other_configs_in_child(local_rank, nstream)
            het = out["other_baseline_configs"].get("heterogeneous_64") if isinstance(out["other_baseline_configs"], dict) else None
            if het and "designs_per_s" in het:
                het["ratio_to_headline"] = het["designs_per_s"] / out["value"]
        if world == 1 and args.cpu_iters != 0:
            cb = cpu_baseline(jobs[0], args.grid_m, infos[0]["iters"], args.cpu_iters)
            out["cpu_baseline"] = cb
            # (no headline ratio: the batch runs 64 designs at once on an O(Mf N) operator, the CPU leg one design on the
            # dense O(Mf N^2) one; the like-for-like pair is the GPU's dense path against it, below)
            if dense_info is not None:
                # like for like: the oracle forms the dense normal matrix, so does the GPU's dense path (one design, one stream)
                out["cpu_baseline"]["gpu_dense_path_designs_per_s"] = 1e3 / dense_info["ms_total"]
                out["cpu_baseline"]["gpu_dense_path_over_cpu"] = 1e3 / dense_info["ms_total"] / cb["value"]
                out["speedup_vs_cpu_baseline"] = out["cpu_baseline"]["gpu_dense_path_over_cpu"]       # dense vs dense, one design each
                # ... and throughput against throughput: the dense path in lock-step units (other_baseline_configs.dense_batch16)
                db = out.get("other_baseline_configs", {}).get("dense_batch16") if isinstance(out.get("other_baseline_configs"), dict) else None
                if db and "designs_per_s" in db:
                    out["cpu_baseline"]["gpu_dense_batch16_designs_per_s"] = db["designs_per_s"]
                    out["cpu_baseline"]["gpu_dense_batch16_over_cpu"] = db["designs_per_s"] / cb["value"]
                if cb["pcost"] is not None:
                    out["cpu_baseline"]["pcost_gpu_vs_cpu"] = [infos[0]["pcost"], cb["pcost"]]
            out["cpu_baseline"]["s_lp_config1"] = s_lp_baseline(mbfir, ctx)
    if world > 1 and not args.no_shard:
        # The row-sharded leg (config 5) runs LAST and under a watchdog: it is the one part of this file that needs every
        # rank to issue the same collectives, so a failure or a stall there must not cost the batch figure -- the line
        # is then printed without "shard" and the processes leave.
        import threading

        def bail():
            # a stall of the sharded leg is a defect, not a result: the batch line is still printed (it was measured),
            # and every rank leaves with a NON-ZERO code so that the launcher and the driver record the failure
            if rank == 0 and out is not None:
                out["shard"] = {"error": "row-sharded leg did not finish within %d s (rank 0 gave up; exit code 3)" % args.shard_timeout}
                print(json.dumps(out), flush=True)
            os._exit(3)
        dog = threading.Timer(args.shard_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            wire_shard(ctx)
            shard_res = shard_leg(args.shard_n, args.shard_grid_m, 1, 1, dense=0)
            # ... and north_star's own design, the dense path with the Gram product on the matrix cores, against the dense
            # path on one GPU: the three predictions of DESIGN section 7 (replicas ~N x, lattice-sharded ~1.3-1.5 x, dense-sharded
            # 5.6-6.2 x at 8 GPUs) are all in the first SCALE record
            if not args.no_shard_dense:
                try:
                    shard_res["dense"] = shard_leg(args.shard_n, args.shard_grid_m, 1, 0, dense=1)
                except Exception as e:                      # noqa: BLE001
                    shard_res["dense"] = {"error": "%s: %s" % (type(e).__name__, e)}
        except Exception as e:                              # noqa: BLE001
            shard_res = {"error": "%s: %s" % (type(e).__name__, e)}
        dog.cancel()
        shard_failed = "error" in shard_res
        if rank == 0:
            out["shard"] = shard_res
    if rank == 0:
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()
    if world > 1 and not args.no_shard and shard_failed:
        sys.exit(3)                                         # the line above carries the batch figure and the error; the failure is not hidden


if __name__ == "__main__":
    main()
