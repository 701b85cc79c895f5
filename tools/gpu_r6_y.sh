#!/bin/bash
# round 6: the OOD rows as a set (plan.ood_rows) -- kernel tests, the CPQ step tests, then A/B against OSRL_OOD_ROWS=0
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6y; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ood_select or row_set or polyak_step_alone or ood_stat" 2>&1 | tail -15 | tee $O/pytest_kernels.log
timeout 1500 python -m pytest tests/test_gpu_train_step.py -x -q -m gpu -k "cpq_small or cpq_c2_full" 2>&1 | tail -5 | tee $O/pytest_step.log
run() {  # cfg label env...
  cfg=$1; lab=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 20 > $O/b.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d.get('executed_gflop_per_step'), d.get('step_frac_executed'), d['config'].get('plan',{}).get('ood_rows'))" $O/b.json $cfg "$lab"
}
for rep in 1 2 3; do
  for cfg in c2 c4; do
    run $cfg ood-rows X=0
    run $cfg all-rows OSRL_LAB=1 OSRL_OOD_ROWS=0
  done
done 2>&1 | tee $O/ab.txt
tail -5 $O/bench.err
