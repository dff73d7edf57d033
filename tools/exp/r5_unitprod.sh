#!/bin/bash
# round 5: the lattice products of a whole unit on the matrix cores, on and off: one unit alone, the bench's batch, under the kernel trace
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5
for mode in 0 4; do
  echo "== MBFIR_UNIT_PRODUCTS=$mode: one unit of 16 alone, then 64 designs on 4 streams"
  MBFIR_UNIT_PRODUCTS=$mode python3 tools/gpu_lanes_one.py 512 16384 16 16 1 2
  MBFIR_UNIT_PRODUCTS=$mode python3 tools/gpu_lanes_one.py 512 16384 64 16 4 3
done
cd /tmp && export TMPDIR=/tmp
for mode in 0 4; do
  MBFIR_UNIT_PRODUCTS=$mode rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5/up$mode -o t -- python3 $GRAFT_REPO_ROOT/tools/gpu_lanes_one.py 512 16384 16 16 1 1 > $GRAFT_REPO_ROOT/gpurun_out/r5/up$mode.log 2>&1
  MBFIR_ROUND=r5 MBFIR_PROFILE_DST=$GRAFT_REPO_ROOT/gpurun_out/r5 python3 -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT/tools'); import rocprof_summary as r
r.kernel_stats('up$mode', 'up${mode}_stats.csv')"
  head -24 $GRAFT_REPO_ROOT/gpurun_out/r5/up${mode}_stats.csv
done
