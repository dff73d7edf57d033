#!/bin/bash
# quick look: bench line + kernel trace of one lock-step unit of 16 headline designs (summary only)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03
mkdir -p $OUT gpurun_out/quick
python3 bench.py --steps 3 --warmup 1 --cpu-iters 0 > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/unit_trace -o unit -- python3 tools/gpu_lanes_one.py 512 16384 16 16 1 1 > gpurun_out/quick/unit_trace.log 2>&1 || exit 1
MBFIR_PROFILE_DST=gpurun_out/quick python3 tools/rocprof_summary.py > gpurun_out/quick/summary.log 2>&1
rm -rf $OUT
python3 -c "
import json; d=json.load(open('gpurun_out/quick/bench.json')); print(d['value'], d['ms_per_step'], d['ipm_iters_per_design'], d['config']['single_design_latency_ms'], d['roofline']['frac'])"
head -30 gpurun_out/quick/r03_unit16_kernel_stats.csv
