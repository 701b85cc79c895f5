#!/bin/bash
# round 6 (second session): steps per graph under the no-join plans (C2 5 / 10, C4 4 / 8), K = 300, two alternating rounds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6nj6; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2; do
  for cfg in c2:5 c2:10 c2:20 c4:4 c4:8 c4:2; do
    c=${cfg%%:*}; n=${cfg##*:}
    timeout 300 python bench.py --config $c --steps-per-graph $n $B > $O/b.json 2> $O/b.err
    echo "$c spg=$n r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
  done
done
