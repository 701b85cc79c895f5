#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4n; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cdt.py -q -x > $O/t1.log 2>&1; tail -5 $O/t1.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
for i in 1 2; do timeout 300 python bench.py --config c5 --no-extras --no-cpu-baseline --no-roofline --no-cold 2>>$O/bench.err | cut -c1-100; done
