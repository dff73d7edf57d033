#!/bin/bash
# round 6: kernel trace of one lock-step unit of 16 headline designs alone (summary only), and of the bench's batch
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MBFIR_ROUND=r06q
OUT=gpurun_out/r06q
mkdir -p $OUT gpurun_out/quick6
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/unit_trace -o unit -- python3 tools/gpu_lanes_one.py 512 16384 16 16 1 1 > gpurun_out/quick6/unit_trace.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/bench_trace -o bench -- python3 bench.py --steps 2 --warmup 1 --cpu-iters 0 --no-other-configs > gpurun_out/quick6/bench_trace.log 2>&1 || exit 1
MBFIR_PROFILE_DST=gpurun_out/quick6 python3 tools/rocprof_summary.py > gpurun_out/quick6/summary.log 2>&1
rm -rf $OUT
ls gpurun_out/quick6
