#!/bin/bash
# round 6 (third session): steps per graph at C4 under the final plan (ood_rows + edges); 300 steps and the driver's K = 20 command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6spg2; rm -rf $O; mkdir -p $O
for r in 1 2 3; do
  for spg in 8 12 16 20; do
    for K in "320 20"; do
      set -- $K
      OSRL_PIPE_STEPS=$spg timeout 300 python bench.py --config c4 --no-cpu-baseline --no-extras --no-roofline --steps $1 --warmup $2 > $O/b.json 2> $O/b.err
      echo "c4 spg=$spg K=$1 r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
    done
  done
done
