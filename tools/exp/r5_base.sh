#!/bin/bash
# round 5: baseline numbers of k_chol_dag (task statistics, alone and under load)
cd "$GRAFT_REPO_ROOT/tools/exp"
mkdir -p ../../gpurun_out/r5
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $CHOL_DEFS chol_dag_exp.hip -o /tmp/chol_dag_exp_r5 2> ../../gpurun_out/r5/build.err || { tail -5 ../../gpurun_out/r5/build.err; exit 1; }
for nl in 8 16; do
  echo "== lanes $nl"
  timeout -k 5 120 /tmp/chol_dag_exp_r5 1024 $nl 4 10 2>&1 | cut -c1-220
done
