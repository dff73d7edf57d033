#!/bin/bash
# round 5 probe: k_chol_dag with ONE workgroup per CU (LDS padded past 80 KB) against the two of the product -- how much of the
# throughput under load comes from the second workgroup of a CU says what a third could add
cd "$GRAFT_REPO_ROOT/tools/exp"
for pad in 0 1000; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DCHOL_PROBE_LDS_PAD=$pad chol_dag_exp.hip -o /tmp/chol_dag_probe_$pad 2>/dev/null || { echo build failed; continue; }
  for nl in 16; do
    echo "== LDS pad $pad doubles, lanes $nl"
    timeout -k 5 120 /tmp/chol_dag_probe_$pad 1024 $nl 4 10 2>&1 | grep -E "SPLIT=4.*per factorisation|words differ|SPLIT=4, 4 units|task statistics|T \(tile|R \(row|MS \(inv|D \(diag" | cut -c1-200
  done
done
