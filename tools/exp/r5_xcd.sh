#!/bin/bash
# round 5: XCD-local hand-offs of k_chol_dag (every lane owned by one XCD: plain stores + L2 atomics) against the write-through form
cd "$GRAFT_REPO_ROOT/tools/exp"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 xcc_probe.hip -o /tmp/xcc_probe 2>/dev/null && { echo "== XCC_ID of the workgroups of a (16, 8) grid, one row per blockIdx.y"; /tmp/xcc_probe; }
bash ./r5_multi.sh "-DCHOL_XCD_LOCAL=1" "-DCHOL_XCD_LOCAL=0"
bash ./r5_sizes.sh "-DCHOL_XCD_LOCAL=1"
