#!/bin/bash
# experiment: persistent workgroups in k_chol_dag (-DCHOL_DAG_PERSIST=<workgroups per lane>) against one workgroup per task
cd "$GRAFT_REPO_ROOT/tools/exp"
mkdir -p ../../gpurun_out/persist
for v in 0 24 32 48 64; do
  if [ $v = 0 ]; then D=""; else D="-DCHOL_DAG_PERSIST=$v"; fi
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $D chol_dag_exp.hip -o /tmp/chol_dag_exp_$v 2> ../../gpurun_out/persist/build_$v.err || { tail -5 ../../gpurun_out/persist/build_$v.err; continue; }
  for nl in 8 16; do
    echo "== persist $v, lanes $nl"
    timeout -k 5 120 /tmp/chol_dag_exp_$v 1024 $nl 4 10 2>&1 | grep -E "per factorisation|words differ|units in flight \(|pivot" | grep -v "task statistics"
  done
done
