#!/bin/bash
# native backtrace of the flaky segfault in the captured data-parallel step (world-1 RCCL job): run the train-step test
# file under rocgdb until it crashes (at most 4 tries)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/segv
for i in 1 2 3 4; do
  timeout 900 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop" -ex run -ex "bt 40" -ex "info sharedlibrary" \
     --args python -m pytest tests/test_gpu_train_step.py -x -q -m gpu $EXTRA > gpurun_out/segv/run$i.txt 2>&1
  if grep -q "SIGSEGV" gpurun_out/segv/run$i.txt; then echo "crashed in run $i"; grep -n "SIGSEGV" -A45 gpurun_out/segv/run$i.txt | head -80; break; else echo "run $i: $(grep -E 'passed|failed' gpurun_out/segv/run$i.txt | tail -1)"; fi
done
