#!/bin/bash
# round 5: kernel statistics of BASELINE config 3 as written, 16 designs in lock-step units of 4 on 4 streams
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5/c3l -o t -- python3 $R/tools/gpu_config3_batch.py 16 4 512 16384 0 > $R/gpurun_out/r5/c3l.log 2>&1
grep "config 3" $R/gpurun_out/r5/c3l.log | cut -c1-120
MBFIR_ROUND=r5 MBFIR_PROFILE_DST=$R/gpurun_out/r5 python3 -c "
import sys; sys.path.insert(0, '$R/tools'); import rocprof_summary as r
r.kernel_stats('c3l', 'c3l_stats.csv')"
head -45 $R/gpurun_out/r5/c3l_stats.csv
rm -rf $R/gpurun_out/r5/c3l
