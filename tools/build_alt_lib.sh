#!/bin/bash
# osrl_amd/lib/libosrl_alt.so = the tree's library with ONE translation unit recompiled under extra flags (A/B of a kernel
# variant through OSRL_LIB):  bash tools/build_alt_lib.sh vae_ns.hip -DOSRL_VAE_NS_PRIO=3
set -e
cd "$(dirname "$0")/.."
python -m osrl_amd.build > /dev/null
src=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -sink-common-insts=false -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16"
b=$(basename $src .hip)
/opt/rocm/bin/hipcc $F "$@" -c osrl_amd/csrc/$src -o /tmp/alt_$b.o
OBJS=$(ls osrl_amd/lib/obj/*.o | grep -v "/$b.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/alt_$b.o $OBJS -L/opt/rocm/lib -lhsa-runtime64 -o osrl_amd/lib/libosrl_alt${SUFFIX}.so
ls -la osrl_amd/lib/libosrl_alt${SUFFIX}.so
