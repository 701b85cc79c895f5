#!/bin/bash
# round 6 (second session): plan.ood_rows (target cost critics on the selected quarter of the N*B rows) inside the no-join graphs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6oodrows2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ood_rows.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
OSRL_OOD_ROWS=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "unjoined or equal_one_step" > $O/pytest_pipe.txt 2>&1; tail -3 $O/pytest_pipe.txt
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in 0 1; do
    for cfg in c2 c4; do
      OSRL_OOD_ROWS=$v timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
      echo "$cfg ood_rows=$v r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
    done
  done
done
tail -2 $O/b.err
