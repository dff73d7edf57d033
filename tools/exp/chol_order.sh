#!/bin/bash
# experiment: ticket order inside a step of k_chol_dag (-DCHOL_DAG_MS_EARLY=1: the inverse-row tasks right behind the diagonal block)
cd "$GRAFT_REPO_ROOT/tools/exp"
for v in 0 1; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DCHOL_DAG_MS_EARLY=$v chol_dag_exp.hip -o /tmp/chol_dag_order_$v 2>/dev/null || { echo build failed; continue; }
  for nl in 8 16; do
    echo "== MS_EARLY $v, lanes $nl"
    timeout -k 5 120 /tmp/chol_dag_order_$v 1024 $nl 4 10 2>&1 | grep -E "SPLIT=4.*per factorisation|words differ|SPLIT=4, 4 units|task statistics|T \(tile|R \(row|MS \(inv|D \(diag" | cut -c1-170
  done
done
