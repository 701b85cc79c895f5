#!/bin/bash
# round 6 (second session): no-join graphs with the actor group's dW + Adam carried to the next step's side branch
# (OSRL_PIPE_ACTOR=side): the main chain is the long one (421 busy vs 392 + 29 idle un-profiled) and nothing on it reads the actor
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6nj8; rm -rf $O; mkdir -p $O
OSRL_PIPE_ACTOR=side timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "unjoined or equal_one_step" > $O/pytest_actor_side.txt 2>&1; tail -3 $O/pytest_actor_side.txt
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2; do
  for v in main:head side:head side:critic side:early; do
    a=${v%%:*}; p=${v##*:}
    OSRL_PIPE_ACTOR=$a OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config c2 $B > $O/b.json 2> $O/b.err
    echo "c2 actor=$a prologue=$p r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
  done
  for v in main:critic side:critic side:head; do
    a=${v%%:*}; p=${v##*:}
    OSRL_PIPE_ACTOR=$a OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config c4 $B > $O/b.json 2> $O/b.err
    echo "c4 actor=$a prologue=$p r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
  done
done
export OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_trace.so
for v in side:head side:critic; do
  a=${v%%:*}; p=${v##*:}
  OSRL_PIPE_ACTOR=$a OSRL_PIPE_PROLOGUE=$p timeout 300 python tools/trace_steps.py c2 5 40 > $O/trace_c2_actor_${a}_$p.txt 2>> $O/b.err
done
