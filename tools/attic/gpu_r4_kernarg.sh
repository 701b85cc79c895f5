#!/bin/bash
# where do launch arguments live?  every config of tools/bench_all.py with device-resident (default) and host-resident kernargs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4k; rm -rf $O; mkdir -p $O
for rep in 1 2; do for v in 1 0; do
  echo "== HIP_FORCE_DEV_KERNARG=$v (rep $rep)"
  HIP_FORCE_DEV_KERNARG=$v timeout 600 python tools/bench_all.py 2>>$O/err.log | grep -v "last stats"
done; done | tee $O/kernarg_ab.txt
