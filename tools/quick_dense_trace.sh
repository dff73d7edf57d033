#!/bin/bash
# dense path, one design: k_gram duration and FETCH_SIZE (re-read factor)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03
mkdir -p $OUT gpurun_out/quick
DENSE="tools/gpu_dense_one.py 512 16384"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/dense_trace -o dense -- python3 $DENSE > gpurun_out/quick/dense_trace.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/dense_pmc_fetch -o dense -- python3 $DENSE > gpurun_out/quick/dense_pmc_fetch.log 2>&1 || exit 1
MBFIR_PROFILE_DST=gpurun_out/quick python3 tools/rocprof_summary.py > gpurun_out/quick/summary.log 2>&1
python3 -c "
import sys; sys.path.insert(0, 'tools')
import rocprof_summary as r
f = r.counter_sums('dense_pmc_fetch')
for k, v in f.items():
    if k.startswith('k_gram'): print(k, 'FETCH_SIZE x2 per launch: %.1f MB (algorithmic 134 MB)' % (v['FETCH_SIZE'] / v['calls'] * 1024 * 2 / 1e6))
"
rm -rf $OUT
head -4 gpurun_out/quick/r03_dense_kernel_stats.csv
grep k_gram gpurun_out/quick/r03_pmc_mfma_dense.csv
