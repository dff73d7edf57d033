#!/bin/bash
# round 5: runtime settings of the HIP stack against the bench's batch and one design alone
cd "$GRAFT_REPO_ROOT"
run() {
  echo "== $*"
  env "$@" python3 tools/gpu_lanes_one.py 512 16384 64 16 4 3 2>&1 | grep designs | cut -c1-110
  env "$@" python3 tools/gpu_lanes_one.py 512 16384 1 1 1 2 2>&1 | grep designs | tail -1 | cut -c1-140
}
run X=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_HIP_KERNARG_COPY_OPT=1
run GPU_FLUSH_ON_EXECUTION=1
run AMD_DIRECT_DISPATCH=0
