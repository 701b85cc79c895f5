#!/bin/bash
# round 6: pipelined steps (engine/pipeline.py) -- bit-equality tests, then steps-per-graph sweeps at C2 / C3 / C4
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pipeline.py "tests/test_gpu_train_step.py::test_random_shape_tuples_match_the_oracle" -x -q --durations=8 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
cp gpurun_out/parity_margins.txt $O/ 2>/dev/null
for cfg in c3 c2 c4; do
  st=200; wu=20; [ $cfg = c3 ] && st=60 && wu=10
  for spg in 1 2 4 10 1 4; do
    timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps $st --warmup $wu --steps-per-graph $spg > $O/b_${cfg}_$spg.json 2>>$O/bench.err
    python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'spg', sys.argv[3], d['value'], d['no_preroll']['value'], d['config'].get('steps_per_graph'))" $O/b_${cfg}_$spg.json $cfg $spg
  done
done 2>&1 | tee $O/sweep.txt
tail -5 $O/bench.err
