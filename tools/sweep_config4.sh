#!/bin/bash
# BASELINE config 4 (256 designs, n=200, m=4096) with the per-step and the single-launch factorisation
cd "$GRAFT_REPO_ROOT"
for split in 1 4; do
  echo "MBFIR_CHOL_SPLIT=$split lanes 32 streams 4"
  MBFIR_CHOL_SPLIT=$split timeout -k 10 200 python3 tools/gpu_lanes_one.py 200 4096 256 32 4 3 || exit 1
done
