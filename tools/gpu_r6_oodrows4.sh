#!/bin/bash
# round 6 (second session): plan.ood_rows at C4 in the no-join graph, with / without the late side start
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6oodrows4; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in 0:0 1:0 1:1; do
    o=${v%%:*}; l=${v##*:}
    OSRL_OOD_ROWS=$o OSRL_OOD_ROWS_LATE=$l timeout 300 python bench.py --config c4 $B > $O/b.json 2> $O/b.err
    echo "c4 ood_rows=$o late=$l r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
