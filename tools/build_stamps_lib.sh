#!/bin/bash
# osrl_amd/lib/libosrl_stamps.so = the tree's library with mlp.hip compiled under -DOSRL_STEP_STAMPS (tools/step_stamps.py)
set -e
cd "$(dirname "$0")/.."
python -m osrl_amd.build > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -sink-common-insts=false -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16"
/opt/rocm/bin/hipcc $F -DOSRL_STEP_STAMPS $EXTRA -c osrl_amd/csrc/mlp.hip -o /tmp/mlp_stamps.o
OBJS=$(ls osrl_amd/lib/obj/*.o | grep -v "/mlp.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/mlp_stamps.o $OBJS -L/opt/rocm/lib -lhsa-runtime64 -o osrl_amd/lib/libosrl_stamps$SUFFIX.so
ls -la osrl_amd/lib/libosrl_stamps$SUFFIX.so
