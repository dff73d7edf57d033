#!/bin/bash
# headline batch (64 designs, n=512, m=16384) over lanes x streams; prints one line per configuration
cd "$GRAFT_REPO_ROOT"
for cfg in "16 4" "13 5" "11 6" "8 8" "16 4"; do
  set -- $cfg
  echo "lanes $1 streams $2"
  timeout -k 10 200 python3 tools/gpu_lanes_one.py 512 16384 64 $1 $2 3 || exit 1
done
