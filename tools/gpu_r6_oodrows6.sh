#!/bin/bash
# round 6 (third session): plan.ood_rows at C2 (late side start, encoder launch on shared-observation tiles): K = 300 and the driver's K = 20 command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6oodrows6; rm -rf $O; mkdir -p $O
for r in 1 2 3 4 5; do
  for v in 0 1; do
    for K in "300 20" "20 5"; do
      set -- $K
      OSRL_OOD_ROWS=$v timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline --steps $1 --warmup $2 > $O/b.json 2> $O/b.err
      echo "c2 ood_rows=$v K=$1 r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
    done
  done
done
