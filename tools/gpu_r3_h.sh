#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3h; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dp_sim.py tests/test_gpu_train_step.py -x -q -m gpu > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -5
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --config c4"
for rep in 1 2; do
  echo -n "[c4 single] "; $B 2>>$O/bench.err | python -c "$P"
  echo -n "[c4 forced dp, round-3 plan] "; OSRL_FORCE_DP=1 $B 2>>$O/bench.err | python -c "$P"
  echo -n "[c4 forced dp, round-1 plan] "; OSRL_FORCE_DP=1 OSRL_DP_PLAN=r1 $B 2>>$O/bench.err | python -c "$P"
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>>$O/bench.err
python -c "import json; d=json.load(open('$O/bench_driver_cmd.json')); print(d['value'], d['ms_per_step'], d['step_frac'], d['step_frac_executed']); print(json.dumps(d['lease'])[:1500]); print(d['roofline'])"
