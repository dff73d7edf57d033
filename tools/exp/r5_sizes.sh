#!/bin/bash
# round 5: bit-identity of the single-launch factorisation against the per-step form over block counts (odd, 32, > 32) and lane counts
cd "$GRAFT_REPO_ROOT/tools/exp"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $1 chol_dag_exp.hip -o /tmp/chol_dag_s 2>/tmp/build.err || { grep error -A3 /tmp/build.err | head; exit 1; }
for cfg in "1024 16 4 6" "1024 8 4 6" "448 16 4 3" "128 5 3 3" "192 7 4 3" "960 3 4 3" "2048 4 3 2" "4096 2 2 1"; do
  echo "== np lanes streams reps: $cfg"
  timeout -k 5 200 /tmp/chol_dag_s $cfg 2>&1 | grep -E "SPLIT=4.*per factorisation|words differ|SPLIT=4, .* units|pivot" | cut -c1-200
done
