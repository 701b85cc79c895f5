#!/bin/bash
# round 6 (second session): N*B-row launches on tiles of shared observations (osrl_rows_t.share0, plan.ood_share) -- kernel tests, the CPQ
# step tests with it on, A/B at C2 / C4
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6share; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "big_rows" > $O/pytest_kernels.txt 2>&1; tail -3 $O/pytest_kernels.txt
export OSRL_LAB=1
OSRL_OOD_SHARE=1 timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_bench_path.py tests/test_gpu_pipeline.py -x -q -k "cpq or c2 or c4" > $O/pytest_step.txt 2>&1; tail -3 $O/pytest_step.txt
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in 0 1; do
    for cfg in c2 c4; do
      OSRL_OOD_SHARE=$v timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
      echo "$cfg ood_share=$v r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
    done
  done
done
tail -3 $O/b.err
OSRL_OOD_SHARE=1 timeout 300 python tools/share_bench.py c2 2>&1 | tail -2 | tee $O/share_bench.txt
OSRL_OOD_SHARE=1 timeout 300 python tools/share_bench.py c4 2>&1 | tail -2 | tee -a $O/share_bench.txt
