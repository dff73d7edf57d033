#!/bin/bash
# headline batch (64 designs, n=512, m=16384) over lanes x streams and unit orders; prints one line per configuration
cd "$GRAFT_REPO_ROOT"
for cfg in "0 16 4" "1 16 4" "1 8 4" "1 8 8" "1 16 3" "0 8 8"; do
  set -- $cfg
  echo "MBFIR_UNIT_ORDER=$1 lanes $2 streams $3"
  MBFIR_UNIT_ORDER=$1 timeout -k 10 200 python3 tools/gpu_lanes_one.py 512 16384 64 $2 $3 2 || exit 1
done
