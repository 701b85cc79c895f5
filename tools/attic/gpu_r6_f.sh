#!/bin/bash
# round 6: the plan-chosen steps per graph -- tests, driver command x 3, other configs; then the PMC re-take (item 3c)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_cmd_$i.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd', d['value'], d['no_preroll']['value'], d['config']['steps_per_graph'], d['roofline']['frac'], {k:(v.get('steps_per_s'), v.get('steps_per_graph')) for k,v in d['other_configs'].items()})" $O/bench_driver_cmd_$i.json
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --steps-per-graph 1 > $O/bench_driver_cmd_spg1_$i.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd, one step per graph', d['value'], d['no_preroll']['value'], d['config']['steps_per_graph'])" $O/bench_driver_cmd_spg1_$i.json
done 2>&1 | tee $O/summary.txt
bash tools/gpu_r6_pmc.sh > $O/pmc.log 2>&1; tail -60 $O/pmc.log
