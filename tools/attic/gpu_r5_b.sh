#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1  # lab switches (OSRL_*) are read only under this (engine/plan.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; rm -rf $O; mkdir -p $O
OSRL_VAE_NS=1 timeout 300 python tools/r5_ns_determinism.py 2>&1 | grep -v amdgpu.ids | tee $O/det1.txt
OSRL_VAE_NS=1 OSRL_ARG_ARENA=0 timeout 300 python tools/r5_ns_determinism.py 2>&1 | grep -v amdgpu.ids | tee $O/det2.txt
