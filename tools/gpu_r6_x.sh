#!/bin/bash
# round 6: split-bf16 GEMM lab (tools/split_gemm_lab.hip) + the f32 kernels of the tree on the same shapes (tools/gemm_lab.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6x; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/split_gemm_lab.hip -o /tmp/split_gemm_lab 2>$O/compile.err || tail -5 $O/compile.err
timeout 600 /tmp/split_gemm_lab $LAB_ARGS 2>&1 | tee $O/split_gemm_lab.txt
