#!/bin/bash
# round 6 (third session): steps per graph at C3 (joined pipelined graphs), 200 steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6spg3; rm -rf $O; mkdir -p $O
for r in 1 2 3; do
  for spg in 10 20 40; do
    OSRL_PIPE_STEPS=$spg timeout 300 python bench.py --config c3 --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 20 > $O/b.json 2> $O/b.err
    echo "c3 spg=$spg r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
