#!/bin/bash
# quick look: kernel trace of the bench's batch (4 units in flight) -> gpurun_out/quick/r03_bench_kernel_stats.csv
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03
mkdir -p $OUT gpurun_out/quick
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o bench -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0 "$@" > gpurun_out/quick/bench_trace.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/unit_trace -o unit -- python3 tools/gpu_lanes_one.py 512 16384 16 16 1 1 > gpurun_out/quick/unit_trace.log 2>&1 || exit 1
MBFIR_PROFILE_DST=gpurun_out/quick python3 tools/rocprof_summary.py > gpurun_out/quick/summary.log 2>&1
python3 tools/trace_concurrency.py $OUT/bench_trace > gpurun_out/quick/concurrency.log 2>&1
rm -rf $OUT
tail -n 1 gpurun_out/quick/bench_trace.log | cut -c1-300
head -24 gpurun_out/quick/r03_bench_kernel_stats.csv
head -12 gpurun_out/quick/r03_unit16_kernel_stats.csv
tail -n 12 gpurun_out/quick/concurrency.log
