#!/bin/bash
# round 6: the fastest of four captures (branch -> hardware-queue mapping): tests, api_path after a pipeline, driver command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6r; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_bench_path.py -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for t in 4 1; do for m in none pipe; do echo "OSRL_CAPTURE_TRIES=$t"; OSRL_LAB=1 OSRL_CAPTURE_TRIES=$t timeout 300 python tools/r6_single_after_pipe.py $m 2>&1 | grep "api_path"; done; done 2>&1 | tee $O/api_path.txt
for i in 1 2 3; do
  for t in 4 1; do
  OSRL_LAB=1 OSRL_CAPTURE_TRIES=$t timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('tries', sys.argv[2], 'driver cmd', d['value'], d['no_preroll']['value'], 'api_path', d['api_path']['steps_per_s'], {k:(v.get('steps_per_s')) for k,v in d['other_configs'].items()})" $O/b.json $t
  done
done 2>&1 | tee $O/summary.txt
