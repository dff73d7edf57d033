"""bench.py contract: one JSON line with the fields the driver reads (small workload so it runs in seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--n", "128", "--grid-m", "4096",
                        "--designs", "8", "--lanes", "4", "--cpu-iters", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "designs/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - d["config"]["designs_per_step_per_rank"]) < 1e-6 * d["value"] * d["ms_per_step"]
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    # the line's roofline is the factorisation at every size (at this toy size the moment build takes about as long, and the
    # entry used to flip with the box to the moment build's "valu" label, which the contract does not know)
    assert rf["kernel"].startswith("k_chol") and 0 < rf["share_of_unit_solve_ms"] < 1
    assert any(o["bound"] == "valu" for o in d["roofline_other"])
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0
    # the untimed legs for BASELINE configs 3 and 4 (full size whatever --n says: 256 designs of n=200; 8 of fir_qp_cvx n=512)
    oc = d["other_baseline_configs"]
    c4, c3 = oc["config4_sweep_256_designs_n200_m4096"], oc["config3_fir_qp_cvx_h1_dualband_n512_m16384"]
    assert c4["solved"] == 256 and c4["designs_per_s"] > 100 and c4["lanes"] == 32
    assert c3["solved"] == 8 and c3["designs_per_s"] > 1 and min(c3["extended_precision_iters"]) > 0 and c3["one_design_status"] == "Solved"
    # (round 5: 16 such designs as lock-step units on the extended-precision path)
    assert c3["batch16_solved"] == 16 and c3["designs_per_s_batch16"] > 1 and max(c3["batch16_lanes_per_unit"]) > 1


def test_bench_full_convergence_cpu_leg_and_distinct_designs():
    """The cpu_baseline leg runs the oracle to convergence (no extrapolation) when asked to, and the timed batch holds
    distinct designs (iteration counts differ)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--n", "100", "--grid-m", "2048",
                        "--designs", "16", "--lanes", "4", "--streams", "2", "--cpu-iters", "-1", "--no-other-configs"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    cb = d["cpu_baseline"]
    assert cb["extrapolated"] is False and "to convergence" in cb["sample"]
    g, c = cb["pcost_gpu_vs_cpu"]
    assert abs(g - c) <= 1e-8 * abs(c)                      # the two legs solved the same design to the same optimum
    assert d["config"]["designs_per_step_per_rank"] == 16 and d["config"]["lanes"] == 4
    assert d["ms_breakdown_lockstep_unit"]["lanes"] == 4 and d["roofline"]["frac"] > 0


def test_row_sharded_bench_as_two_processes_over_gloo():
    """`bench.py --gpus 2 --mode shard --backend gloo`: two ranks launched as child processes (before this process's
    children touch the GPU), sharing the one GPU, the solver's reductions going through the host hook.  Exercises the
    multi-process path end to end: rendezvous, partition, every collective of the sharded iteration, rank-0 output.
    The taps of the sharded solve equal the unsharded solve's (checked through the objective and the iteration count
    in the JSON; tests/test_shard_gpu.py compares taps)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "shard", "--backend", "gloo",
                        "--steps", "1", "--warmup", "0", "--taps", "64", "--grid-m", "1024"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["shard"]["collectives_per_iteration"] > 0 and "hook" in d["shard"]["reductions"]
    # the measured split into the replicated factorisation and the shardable row work, and the bound it implies
    ps = d["shard"]["phase_split_ms"]
    assert ps["replicated_factorisation"] > 0 and ps["shardable_row_work_one_gpu"] > 0 and ps["sharded_solve_total"] > 0
    assert 1.0 <= d["shard"]["amdahl_bound_speedup_at_this_n_gpus"] <= 2.0


def test_two_rank_batch_bench_carries_the_shard_leg():
    """`bench.py --gpus 2` in its default (batch, weak-scaling) mode as the driver launches it, over gloo on one GPU: the
    line carries the whole-job designs/s of both ranks and, under "shard", the row-sharded strong-scaling leg."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "1", "--taps", "64", "--grid-m", "1024", "--designs", "8", "--lanes", "4", "--streams", "2",
                        "--shard-n", "64", "--shard-grid-m", "1024", "--cpu-iters", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 2 * 8) < 1e-6 * d["value"] * d["ms_per_step"]      # both ranks' designs
    assert "error" not in d["shard"] and d["shard"]["scaling"] == "strong" and d["shard"]["value"] > 0 and d["shard"]["mode"] == "lattice"
    # round 5: the same leg on the dense path (north_star's design: Gram product on the matrix cores, packed lower triangle all-reduced)
    dd = d["shard"]["dense"]
    assert "error" not in dd and dd["mode"] == "dense" and dd["value"] > 0 and dd["collective_MB_per_iteration"] > 0


def test_plain_bench_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment (the form of the driver's N = 1 command): bench.py
    spawns its two ranks itself and passes rank 0's one JSON line and the child's exit code through."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                        "--warmup", "1", "--taps", "64", "--grid-m", "1024", "--designs", "8", "--lanes", "4", "--streams", "2",
                        "--no-shard", "--cpu-iters", "0", "--no-other-configs"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
