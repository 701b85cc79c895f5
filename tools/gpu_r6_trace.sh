#!/bin/bash
# round 6 (second session): un-profiled kernel start stamps (csrc/trace.h) of the joined and the no-join C2 / C4 graphs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1 OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_trace.so
O=$GRAFT_REPO_ROOT/gpurun_out/r6trace; rm -rf $O; mkdir -p $O
for v in next:head; do
  d=${v%%:*}; p=${v##*:}
  OSRL_PIPE_DUAL=$d OSRL_PIPE_PROLOGUE=$p timeout 300 python tools/trace_steps.py c2 5 60 > $O/trace_c2_${d}_${p}.txt 2> $O/trace_c2_${d}_${p}.err
  head -3 $O/trace_c2_${d}_${p}.txt; tail -2 $O/trace_c2_${d}_${p}.err
  OSRL_PIPE_DUAL=$d OSRL_PIPE_PROLOGUE=$p timeout 300 python tools/trace_steps.py c4 4 60 > $O/trace_c4_${d}_${p}.txt 2> $O/trace_c4_${d}_${p}.err
  head -3 $O/trace_c4_${d}_${p}.txt
done
export OSRL_LIB=
unset OSRL_LIB
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2; do
  for v in main:early next:head; do
    d=${v%%:*}; p=${v##*:}
    for cfg in c2:5 c4:4; do
      c=${cfg%%:*}; n=${cfg##*:}
      [ $c = c4 ] && [ $d = main ] && d=next && p=critic
      OSRL_PIPE_DUAL=$d OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config $c --steps-per-graph $n $B > $O/b.json 2> $O/b.err
      echo "$c spg=$n dual=$d prologue=$p r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
      d=${v%%:*}; p=${v##*:}
    done
  done
done
