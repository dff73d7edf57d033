#!/bin/bash
# round 5: kernel statistics of one unit of 16 alone with the launch fusions on and off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r5
for mode in 0 1; do
  MBFIR_FUSE=$mode rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5/fu$mode -o t -- python3 $R/tools/gpu_lanes_one.py 512 16384 16 16 1 1 > $R/gpurun_out/r5/fu$mode.log 2>&1
  MBFIR_ROUND=r5 MBFIR_PROFILE_DST=$R/gpurun_out/r5 python3 -c "
import sys; sys.path.insert(0, '$R/tools'); import rocprof_summary as r
r.kernel_stats('fu$mode', 'fu${mode}_stats.csv')"
  echo "== MBFIR_FUSE=$mode"; head -26 $R/gpurun_out/r5/fu${mode}_stats.csv
  rm -rf $R/gpurun_out/r5/fu$mode
done
