#!/bin/bash
# osrl_amd/lib/libosrl_trace.so = the tree's library with every translation unit compiled under -DOSRL_TRACE (csrc/trace.h:
# kernel START stamps of an un-profiled run; tools/trace_steps.py).  Lab only -- the product library never defines it.
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -sink-common-insts=false -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=16 -DOSRL_TRACE"
D=/tmp/osrl_trace_obj; mkdir -p $D
pids=()
for s in osrl_amd/csrc/*.hip; do
  b=$(basename $s .hip); X=""
  [ $b = cdt ] && X="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc $F $X -c $s -o $D/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -L/opt/rocm/lib -lhsa-runtime64 -o osrl_amd/lib/libosrl_trace.so
ls -la osrl_amd/lib/libosrl_trace.so
