#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py -q -x -k "seeded or superseded or bcql or bearl" > $O/t1.log 2>&1; tail -8 $O/t1.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --no-cold --config c3"
for rep in 1 2 3; do for v in 0 1; do echo -n "c3 OSRL_SEEDS=$v "; env OSRL_SEEDS=$v $B 2>>$O/bench.err | cut -c1-60; done; done
OSRL_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_force_dp.json 2>> $O/bench.err
python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('force_dp', d['value'], d.get('collectives_in_step'), (d.get('other_configs') or {}).get('c4'))" $O/bench_force_dp.json
