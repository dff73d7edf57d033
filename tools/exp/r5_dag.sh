#!/bin/bash
# round 5: chol_dag_exp (bit-identity against the per-step form, time alone and under load, task statistics with phase stamps)
# usage: r5_dag.sh "<extra -D flags>" "<lane counts>"
cd "$GRAFT_REPO_ROOT/tools/exp"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $1 chol_dag_exp.hip -o /tmp/chol_dag_r5 2>/tmp/build.err || { grep -E "error" -A5 /tmp/build.err | head -40; exit 1; }
for nl in ${2:-16 8}; do
  echo "== flags [$1] lanes $nl"
  timeout -k 5 120 /tmp/chol_dag_r5 1024 $nl 4 10 2>&1 | grep -v "lane 0 diagonal" | cut -c1-400
done
