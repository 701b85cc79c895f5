#!/bin/bash
# round 3: device-resident argument blocks.  Tests of the train step, then the 2x2 A/B (arena x kernarg placement)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'
for rep in 1 2; do
  for v in "HIP_FORCE_DEV_KERNARG=1 OSRL_ARG_ARENA=1" "HIP_FORCE_DEV_KERNARG=1 OSRL_ARG_ARENA=0" "HIP_FORCE_DEV_KERNARG=0 OSRL_ARG_ARENA=1" "HIP_FORCE_DEV_KERNARG=0 OSRL_ARG_ARENA=0"; do
    echo -n "[$v] "; env $v $B 2>>$O/bench.err | python -c "$P"
  done
done
for c in c3 c4 c5 c1; do
  for v in "HIP_FORCE_DEV_KERNARG=1 OSRL_ARG_ARENA=1" "HIP_FORCE_DEV_KERNARG=0 OSRL_ARG_ARENA=1" "HIP_FORCE_DEV_KERNARG=0 OSRL_ARG_ARENA=0"; do
    echo -n "[$c $v] "; env $v $B --config $c --steps 100 2>>$O/bench.err | python -c "$P"
  done
done
