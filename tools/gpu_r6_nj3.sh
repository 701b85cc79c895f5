#!/bin/bash
# round 6 (second session): no-join pipelined CPQ graphs, the next prologue in front of the critic phase (covered by ev_critic)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6nj3; rm -rf $O; mkdir -p $O
OSRL_PIPE_DUAL=next OSRL_PIPE_PROLOGUE=critic timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "cpq or c2 or c4" > $O/pytest_next_critic.txt 2>&1; tail -3 $O/pytest_next_critic.txt
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2; do
  for v in main:early next:early next:critic main:critic; do
    d=${v%%:*}; p=${v##*:}
    for cfg in c2:5 c4:4; do
      c=${cfg%%:*}; n=${cfg##*:}
      OSRL_PIPE_DUAL=$d OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config $c --steps-per-graph $n $B > $O/b_${c}_${d}_${p}_$r.json 2> $O/b_${c}_${d}_${p}_$r.err
      echo "$c spg=$n dual=$d prologue=$p r$r $(python -c "import json,sys; d=json.loads(open('$O/b_${c}_${d}_${p}_$r.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
    done
  done
done
