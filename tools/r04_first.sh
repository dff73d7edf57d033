#!/bin/bash
# Round 4, first GPU call: the suite on the new build, a baseline bench line, then the profiled 8-stream extended-precision
# batch ONCE with the fault-time module map armed (tools/r04_first.sh through gpurun from the repo root).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r04
mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_switches_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "eight_contexts or heterogeneous or lost_in_launch or units_start or lock_step" > $OUT/gputest1.log 2>&1 || { tail -40 $OUT/gputest1.log; exit 1; }
tail -2 $OUT/gputest1.log
timeout -k 10 600 python3 tools/gpu_fuzz_lockstep.py 0 40 4 0 edges > $OUT/fuzz_edges1.log 2>&1 || { tail -20 $OUT/fuzz_edges1.log; exit 1; }
tail -8 $OUT/fuzz_edges1.log
timeout -k 10 600 python3 bench.py --steps 5 --warmup 1 > $OUT/bench1.json 2> $OUT/bench1.err || { tail -20 $OUT/bench1.err; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/bench1.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('other_baseline_configs'))"
export MBFIR_FAULT_MAPS=$GRAFT_REPO_ROOT/$OUT/fault_maps.txt
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/c3_trace -o c3 -- python3 tools/gpu_config3_batch.py 8 8 > $OUT/c3_trace.log 2>&1
echo "profiled 8-stream config-3 batch: exit $?"
grep "config 3" $OUT/c3_trace.log
ls -la $OUT/fault_maps.txt 2>/dev/null
rm -rf $OUT/c3_trace
