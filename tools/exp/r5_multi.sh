#!/bin/bash
# round 5: several builds of chol_dag_exp in one call: r5_multi.sh "<flags A>" "<flags B>" ... (lanes 16 and 8 each)
cd "$GRAFT_REPO_ROOT/tools/exp"
for f in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $f chol_dag_exp.hip -o /tmp/chol_dag_m 2>/tmp/build.err || { echo "build failed [$f]"; grep error -A3 /tmp/build.err | head; continue; }
  for nl in 16 8; do
    echo "== flags [$f] lanes $nl"
    timeout -k 5 120 /tmp/chol_dag_m 1024 $nl 4 8 2>&1 | grep -E "SPLIT=4.*per factorisation|words differ|SPLIT=4, 4 units|task statistics|tasks " | cut -c1-330
  done
done
