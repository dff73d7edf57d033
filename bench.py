#!/usr/bin/env python3
"""Headline benchmark: FIR designs/sec (and IPM iterations/sec) of the arbitrary-phase SOCP designer
at n=512 taps, m=16384 grid points (BASELINE.json metric), on N GPUs of one node.

A "step" is one batch of `--streams` (default 4) independent designs through the C ABI
(mbfir_solve_batch -> mbfir_ap_solve: host assembly, every IPM iteration on the GPU, spectral
factorisation), all of the S-C13 bSSFP spec (bSSFP_pulse_sb_mb.m:9-52) in the fixed-duration regime,
obj=0.1, Peak=1e-3 -- the shape of the reference's outer loops (bisection probes, parameter sweeps).
The designs of a batch run on separate HIP streams, so the latency-bound phases of one (Cholesky
panels, reductions, the per-iteration host check) overlap the others; the single-design latency is
reported beside the throughput.  Inputs are a few dozen doubles, so "inputs resident in HBM" is
trivially true; the timed region includes the PCIe hand-over of the specs and of the taps.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the default mode gives every
rank its own batches (independent designs shard with no data-path collective, SURVEY 8e "replicas
only"), so scaling is weak and value = N * streams * K / max-over-ranks time.  --mode shard splits
the frequency rows of ONE design over the ranks (RCCL all-reduce per iteration, strong scaling).

One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F100 = [-0.241994, -0.233994, -0.152431, -0.144431, -0.083851, -0.075851, -0.052798, -0.044798, -0.004, 0.004]
A_C13 = [0.0] * 8 + [0.500125, 0.500125]
D_C13 = [0.00250001] * 4 + [0.00866503]
PEAK_FP64_MATRIX_TF = 78.6     # AMD's public MI355X fp64 matrix figure; the local hardware guide lists none
# HBM-side bytes of one k_chol_step launch at np = 1024 from the PMC passes of this round
# (profiles/r01d_pmc_fetch_write_per_kernel.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate
# runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B/lane streaming reads on gfx950):
# 2 x 3475.8 KB + 2620.7 KB.  Algorithmic: the tiles a panel step touches (~96 KB read per updated tile,
# 32 KB written), 8.7 MB read + 2.6 MB written on average -- no re-read excess.
CHOL_STEP_TRAFFIC_BYTES_NP1024 = (2 * 3475.8 + 2620.7) * 1024


def workload(n):
    s = 100.0 / n                     # fixed duration T = 4 ms: fs = n/T, band edges scale with 100/n
    return [x * s for x in F100], A_C13, D_C13


def cpu_baseline(n, grid_m, gpu_iters, iters_cpu):
    """The oracle (NumPy/OpenBLAS port of the same algorithm) on this box's host cores, on a bounded
    sample: assembly + the first `iters_cpu` IPM iterations; per-iteration cost is constant, so the
    per-design time is extrapolated to the iteration count the GPU run needed."""
    from oracle import assemble, conic_ipm
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    f, a, d = workload(n)
    t0 = time.perf_counter()
    P = assemble.assemble_fir_ap_cvx(n, f, a, d, 0.1, 1e-3, grid_m)
    t1 = time.perf_counter()
    conic_ipm.solve(P["c"], P["G"], P["h"], P["l"], P["nq3"], P["big"], max_iter=iters_cpu)
    t2 = time.perf_counter()
    t_iter = (t2 - t1) / (iters_cpu + 1)            # the initial point costs one factorisation + solve
    t_design = (t1 - t0) + t_iter * (gpu_iters + 1)
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": 1.0 / t_design, "unit": "designs/s", "cores": int(cores), "kind": "port",
            "sample": "oracle (NumPy/OpenBLAS dense conic IPM, same algorithm): assembly %.1f s + first %d of %d IPM "
                      "iterations (%.2f s each), extrapolated to the full design" % (t1 - t0, iters_cpu, gpu_iters, t_iter),
            "s_per_iteration": t_iter, "iters_per_s": 1.0 / t_iter}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--grid-m", type=int, default=16384)
    ap.add_argument("--mode", choices=["batch", "shard"], default="batch")
    ap.add_argument("--cpu-iters", type=int, default=8, help="oracle iterations for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--streams", type=int, default=4, help="independent designs in flight per GPU (contexts / HIP streams); "
                    "a step is one batch of that many designs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a box with fewer GPUs than ranks: ranks then share devices)")
    ap.add_argument("--dense", action="store_true", help="materialised trig matrix + dense MFMA Gram (opts.dense_trig) "
                    "instead of the default lattice (matrix-free) mode")
    args = ap.parse_args()

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
    f, a, d = workload(args.n)
    shard = args.mode == "shard" and world > 1
    nstream = 1 if shard else max(1, args.streams)
    ctxs = [mbfir.Context(local_rank) for _ in range(nstream)]
    ctx = ctxs[0]
    if shard:
        # ONE design, its frequency rows split over the ranks; per iteration RCCL all-reduces the moments of the
        # normal matrix (dense path: the matrix), every G'v / preconditioner application and the step / residual
        # scalars (mbfir_set_allreduce hook)
        if args.backend == "nccl":
            ctx.set_allreduce(mbfir.make_torch_allreduce())
        else:                                            # gloo rehearsal: stage the device buffer through the host
            import torch

            def hook(ptr, count, op, _t=torch):
                import torch.distributed as dist
                t = mbfir.device_tensor(ptr, count)
                c = t.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
                t.copy_(c)
                _t.cuda.synchronize()
                return 0
            ctx.set_allreduce(hook)
        opts = mbfir.make_opts(grid_m=args.grid_m, shard_rank=rank, shard_size=world, dense_trig=int(args.dense))
    else:
        opts = mbfir.make_opts(grid_m=args.grid_m, dense_trig=int(args.dense))

    jobs = [("fir_ap_cvx", (args.n, f, a, d, 0.1, 1e-3))] * nstream

    def step():
        if shard:
            res = [mbfir.fir_ap_cvx(args.n, f, a, d, 0.1, 1e-3, opts=opts, ctx=ctx, info=True)]
        else:
            res = mbfir.solve_batch(jobs, opts=opts, ctxs=ctxs, info=True)      # mbfir_solve_batch: nstream designs in flight
        for h, status, info in res:
            if status != "Solved":
                raise RuntimeError("benchmark design did not solve: %r" % (info,))
        return [info for _, _, info in res]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    infos = [i for _ in range(args.steps) for i in step()]
    fence()
    elapsed = time.perf_counter() - t0
    ndesign = args.steps * nstream            # designs this rank completed in the timed region
    # single-design latency (one stream, nothing else on the GPU), outside the timed region
    t1 = time.perf_counter()
    _, _, solo = mbfir.fir_ap_cvx(args.n, f, a, d, 0.1, 1e-3, opts=opts, ctx=ctx, info=True)
    torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t1) * 1e3
    # the north-star kernel A'DA on the matrix cores is the dense path's (opts.dense_trig); the default
    # lattice path replaces it by moments.  One dense design, outside the timed region, keeps its live
    # MFMA rate in the record.
    dense_info = None
    if not args.dense and not shard:
        _, st_d, dense_info = mbfir.fir_ap_cvx(args.n, f, a, d, 0.1, 1e-3, ctx=ctx, info=True,
                                               opts=mbfir.make_opts(grid_m=args.grid_m, dense_trig=1))
        if st_d != "Solved":
            dense_info = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        iters = sum(i["iters"] for i in infos)
        # kernel rooflines come from the single-stream pass (`solo`: one design alone on the GPU, same build):
        # HIP events around a stream's launches also see the other streams' kernels when several designs are
        # in flight, while rocprofv3 reports pure kernel durations -- alone on the GPU the two agree
        # (profiles/*_streams1_kernel_stats.csv).  The per-design breakdown below stays that of the timed region.
        builds = solo["builds"]
        gram_ms = solo["ms_gram"]
        chol_ms = solo["ms_chol"]
        chol_launches = solo["chol_launches"]
        lattice = bool(infos[0]["lattice"])
        try:
            peak_mfma, peak_valu = mbfir.mfma_peak(ctx)
        except Exception:
            peak_mfma = peak_valu = float("nan")
        # k_chol_step: one launch per 64-wide panel step of the Cholesky + triangular inverse; the
        # algorithmic flop of a build (2/3 np^3) spread over its np/64 + 1 launches
        chol_flop_per_launch = infos[0]["chol_flop"] * builds / max(1, chol_launches)
        chol_avg_ms = chol_ms / max(1, chol_launches)
        chol_ach = chol_flop_per_launch / (chol_avg_ms * 1e-3) / 1e12 if chol_ms > 0 else 0.0
        roof_chol = {"kernel": "k_chol_step (blocked Cholesky + triangular inverse, one launch per 64-wide panel; tile "
                               "products on v_mfma_f64_16x16x4_f64)", "bound": "mfma", "achieved": chol_ach,
                     "peak": PEAK_FP64_MATRIX_TF, "unit": "TFLOP/s", "frac": chol_ach / PEAK_FP64_MATRIX_TF,
                     "traffic": CHOL_STEP_TRAFFIC_BYTES_NP1024 if infos[0]["n_unknowns"] == 1024 else None,
                     "traffic_unit": "bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01d_pmc_fetch_write_per_kernel.csv)",
                     "flop_per_launch": chol_flop_per_launch, "launches": chol_launches, "avg_launch_ms": chol_avg_ms,
                     "note": "dependency-chain bound, not throughput bound: 1024 sequential pivots per build (single-wave "
                             "register elimination of 16-column slabs, ~0.1 us per pivot) plus 17 launch boundaries; measured on "
                             "the single-stream pass; see DESIGN.md",
                     "peak_source": "AMD public MI355X fp64 matrix figure (not in MI355X_MICROARCH.md)",
                     "peak_measured_mfma_f64": peak_mfma, "peak_measured_valu_f64": peak_valu}
        # normal-matrix products: dense mode = k_gram on the matrix cores; lattice mode = moment recurrences on the VALU
        gram_flop = infos[0]["gram_flop"]
        if lattice:
            gram_ach = gram_flop * builds / (gram_ms * 1e-3) / 1e12 if gram_ms > 0 else 0.0
            roof_gram = {"kernel": "k_trig_moments + fold + k_assemble_H_lat (normal matrix from trigonometric moments, fp64 VALU)",
                         "bound": "valu", "achieved": gram_ach, "peak": PEAK_FP64_MATRIX_TF, "unit": "TFLOP/s",
                         "frac": gram_ach / PEAK_FP64_MATRIX_TF, "traffic": None, "flop_per_build": gram_flop, "builds": builds,
                         "avg_build_ms": gram_ms / max(1, builds),
                         "note": "replaces the dense Gram products (17.2 GFLOP per build on the matrix cores) by "
                                 "%.2f GFLOP of recurrences; peak = fp64 vector peak (same figure as the matrix peak)" % (gram_flop / 1e9)}
        else:
            launches = solo["gram_launches"]
            flop_per_launch = gram_flop / max(1, infos[0]["gram_launches"] // max(1, infos[0]["builds"]))
            gram_ach = flop_per_launch / (gram_ms / max(1, launches) * 1e-3) / 1e12 if gram_ms > 0 else 0.0
            roof_gram = {"kernel": "k_gram (A' D A, v_mfma_f64_16x16x4_f64)", "bound": "mfma", "achieved": gram_ach,
                         "peak": PEAK_FP64_MATRIX_TF, "unit": "TFLOP/s", "frac": gram_ach / PEAK_FP64_MATRIX_TF,
                         "traffic": None, "flop_per_launch": flop_per_launch, "launches": launches,
                         "avg_launch_ms": gram_ms / max(1, launches),
                         "peak_source": "AMD public MI355X fp64 matrix figure (not in MI355X_MICROARCH.md)",
                         "peak_measured_mfma_f64": peak_mfma, "peak_measured_valu_f64": peak_valu}
        dominant, other = (roof_chol, roof_gram) if chol_ms >= gram_ms else (roof_gram, roof_chol)
        others = [other]
        if dense_info is not None and dense_info["gram_launches"] > 0:
            fl = dense_info["gram_flop"] / max(1, dense_info["gram_launches"] // max(1, dense_info["builds"]))
            avg = dense_info["ms_gram"] / dense_info["gram_launches"]
            ach_d = fl / (avg * 1e-3) / 1e12
            others.append({"kernel": "k_gram (A' D A, v_mfma_f64_16x16x4_f64) -- dense path (opts.dense_trig=1), one design "
                                     "after the timed region", "bound": "mfma", "achieved": ach_d, "peak": PEAK_FP64_MATRIX_TF,
                           "unit": "TFLOP/s", "frac": ach_d / PEAK_FP64_MATRIX_TF, "traffic": None, "flop_per_launch": fl,
                           "launches": dense_info["gram_launches"], "avg_launch_ms": avg,
                           "dense_design_ms": dense_info["ms_total"]})
        out = {
            "metric": "FIR designs/sec, n=%d taps m=%d arbitrary-phase SOCP (fir_ap_cvx form)" % (args.n, args.grid_m),
            "value": (1 if shard else world) * ndesign / elapsed, "unit": "designs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "S-C13 bSSFP 5-band spec, fixed-duration regime, fir_ap_cvx(n=%d, obj=0.1, Peak=1e-3), "
                                   "grid_m=%d (+10 band edges); a step is one batch of %d independent designs in flight per rank "
                                   "(mbfir_solve_batch, one HIP stream each)" % (args.n, args.grid_m, nstream),
                       "designs_per_step_per_rank": nstream, "single_design_latency_ms": latency_ms,
                       "n_taps": args.n, "grid_m": args.grid_m, "unknowns": infos[0]["n_unknowns"], "rows": infos[0]["n_rows"],
                       "mode": args.mode, "trig": "lattice (matrix-free)" if lattice else "dense (materialised trig matrix, MFMA Gram)",
                       "parallelism": ("frequency rows of one design sharded x%d, RCCL all-reduce per iteration" % world) if shard
                       else "independent designs x%d" % world},
            "ipm_iters_per_design": iters / ndesign,
            "ipm_iters_per_s": (1 if shard else world) * iters / elapsed,
            "ms_breakdown_per_design": {"assemble": sum(i["ms_assemble"] for i in infos) / ndesign,
                                        "solve": sum(i["ms_solve"] for i in infos) / ndesign,
                                        "normal_matrix": sum(i["ms_gram"] for i in infos) / ndesign,
                                        "cholesky_inverse": sum(i["ms_chol"] for i in infos) / ndesign,
                                        "spectral_factor": sum(i["ms_post"] for i in infos) / ndesign,
                                        "note": "per-stream device/host times while %d designs share the GPU" % nstream},
            "ms_breakdown_single_stream": {"assemble": solo["ms_assemble"], "solve": solo["ms_solve"], "normal_matrix": solo["ms_gram"],
                                           "cholesky_inverse": solo["ms_chol"], "spectral_factor": solo["ms_post"]},
            "roofline": dominant,
            "roofline_other": others,
        }
        if world == 1 and args.cpu_iters > 0:
            out["cpu_baseline"] = cpu_baseline(args.n, args.grid_m, int(round(iters / ndesign)), args.cpu_iters)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
