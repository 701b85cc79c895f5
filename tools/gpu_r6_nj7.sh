#!/bin/bash
# round 6 (second session): the driver's K = 20 / W = 5 command under the no-join C2 plan at 5 / 10 / 20 steps per graph
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6nj7; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 5"
for r in 1 2 3 4 5 6; do
  for n in 5 10 20; do
    timeout 300 python bench.py --config c2 --steps-per-graph $n $B > $O/b.json 2> $O/b.err
    echo "c2 K=20 spg=$n r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
  done
done
