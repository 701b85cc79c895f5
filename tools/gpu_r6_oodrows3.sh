#!/bin/bash
# round 6 (second session): plan.ood_rows at C2 in the no-join graph under the three prologue placements; un-profiled stamps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6oodrows3; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2; do
  for v in 0:head 1:head 1:critic; do
    o=${v%%:*}; p=${v##*:}
    OSRL_OOD_ROWS_LATE=1 OSRL_OOD_ROWS=$o OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config c2 $B > $O/b.json 2> $O/b.err
    echo "c2 ood_rows=$o prologue=$p r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
OSRL_OOD_ROWS_LATE=1 OSRL_OOD_ROWS=1 OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_trace.so timeout 300 python tools/trace_steps.py c2 5 40 > $O/trace_c2_oodrows.txt 2>> $O/b.err
head -2 $O/trace_c2_oodrows.txt
