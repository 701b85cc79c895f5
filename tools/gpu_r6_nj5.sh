#!/bin/bash
# round 6 (second session): how often does the driver's K = 20 command land a stall in a timed region -- joined vs no-join graph
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6nj5; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 5"
for r in $(seq 1 12); do
  for v in main:early next:head; do
    d=${v%%:*}; p=${v##*:}
    OSRL_PIPE_DUAL=$d OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config c2 $B > $O/b.json 2> $O/b.err
    echo "c2 dual=$d prologue=$p r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
  done
done
