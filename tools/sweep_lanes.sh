#!/bin/bash
# headline batch (64 designs, n=512, m=16384) over lanes x streams; prints one line per configuration
cd "$GRAFT_REPO_ROOT"
export MBFIR_MAX_LANES=64
for cfg in "8 4" "16 4" "32 2" "64 1" "32 1" "16 2"; do
  set -- $cfg
  echo "lanes $1 streams $2"
  timeout -k 10 200 python3 tools/gpu_lanes_one.py 512 16384 64 $1 $2 2 || exit 1
done
