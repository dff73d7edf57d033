#!/bin/bash
# round 5: lattice moments of a whole lock-step unit on the matrix cores against the per-lane kernel with the same arithmetic
cd "$GRAFT_REPO_ROOT/tools/exp"
mkdir -p ../../gpurun_out/r5
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -w trig_mf_exp.hip -o /tmp/trig_mf_exp 2> ../../gpurun_out/r5/trigmf_build.err || { tail -5 ../../gpurun_out/r5/trigmf_build.err; exit 1; }
timeout -k 5 120 /tmp/trig_mf_exp 50 2>&1 | tee ../../gpurun_out/r5/trigmf.log
