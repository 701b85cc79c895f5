#!/bin/bash
# round 6 (third session): where the side branch's second half starts under plan.ood_rows -- behind the cost critics' dW (shipped) / Adam;
# C4's prologue placement under the final plan
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6late2; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in c2:0:auto c2:1:auto c4:0:auto c4:1:auto c4:0:early c4:0:head; do
    IFS=: read cfg l p <<< "$v"
    OSRL_OOD_ROWS_LATE2=$l OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
    echo "$cfg late2=$l prologue=$p r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
tail -n 2 $O/b.err
