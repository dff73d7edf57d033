#!/bin/bash
# Round 4, second GPU call: the whole GPU suite on the heterogeneous-unit build, the discriminating profiled run for round 3's
# crash (8 streams with GPU_MAX_HW_QUEUES=8: no two streams share an HSA queue), the fir_ap probe timing.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/gputest2.log 2>&1 || { tail -40 $OUT/gputest2.log; exit 1; }
tail -2 $OUT/gputest2.log
timeout -k 10 300 python3 tools/gpu_search_probes.py 260 > $OUT/search_probes.log 2>&1 || { tail -20 $OUT/search_probes.log; exit 1; }
cat $OUT/search_probes.log
export MBFIR_FAULT_MAPS=$GRAFT_REPO_ROOT/$OUT/fault_maps_q8.txt
GPU_MAX_HW_QUEUES=8 timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/c3_trace -o c3 -- python3 tools/gpu_config3_batch.py 8 8 > $OUT/c3_trace_q8.log 2>&1
echo "profiled 8-stream config-3 batch with GPU_MAX_HW_QUEUES=8: exit $?"
grep "config 3" $OUT/c3_trace_q8.log
ls -la $OUT/fault_maps_q8.txt 2>/dev/null
rm -rf $OUT/c3_trace
