#!/bin/bash
# round 6 (second session): C2, no-join graph with the next prologue at the head of the side branch against the joined form:
# four alternating rounds at K = 300 and the driver's K = 20 / W = 5 command (headline fields only)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6nj4; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline"
for r in 1 2 3 4; do
  for v in main:early next:head; do
    d=${v%%:*}; p=${v##*:}
    for k in "300 20" "20 5"; do
      set -- $k
      OSRL_PIPE_DUAL=$d OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config c2 $B --steps $1 --warmup $2 > $O/b.json 2> $O/b.err
      echo "c2 dual=$d prologue=$p K=$1 r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
    done
  done
done
