#!/bin/bash
# headline batch (64 designs, n=512, m=16384) over lanes x streams and Cholesky forms; prints one line per configuration
cd "$GRAFT_REPO_ROOT"
for cfg in "1 8 4" "4 8 4" "4 16 2" "4 16 4" "4 32 2" "4 8 8" "4 16 3"; do
  set -- $cfg
  echo "MBFIR_CHOL_SPLIT=$1 lanes $2 streams $3"
  MBFIR_CHOL_SPLIT=$1 timeout -k 10 200 python3 tools/gpu_lanes_one.py 512 16384 64 $2 $3 2 || exit 1
done
