#!/bin/bash
# round 5 probe: what bounds k_chol_dag under load?  Timing-only builds (wrong results): no MFMA k-loops, no substitutions, both
cd "$GRAFT_REPO_ROOT/tools/exp"
for f in "" "-DCHOL_EXP_NO_MFMA" "-DCHOL_EXP_NO_SUBST" "-DCHOL_EXP_NO_MFMA -DCHOL_EXP_NO_SUBST"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 $f chol_dag_exp.hip -o /tmp/chol_dag_p3 2>/dev/null || { echo build failed; continue; }
  echo "== flags [$f]"
  timeout -k 5 120 /tmp/chol_dag_p3 1024 16 4 6 2>&1 | grep -E "SPLIT=4.*per factorisation|SPLIT=4, 4 units|task statistics|tasks " | cut -c1-330
done
