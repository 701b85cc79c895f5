#!/bin/bash
# round 6 (second session): pipelined CPQ graphs WITHOUT a join between their steps (OSRL_PIPE_DUAL=next: step k's dual
# step at the head of step k+1's side branch; the main chain waits for its prologue's event only) -- bit-equality tests,
# A/B against the joined form at C2 / C4, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6nj; rm -rf $O; mkdir -p $O
OSRL_PIPE_DUAL=next timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "cpq or c2 or c4" > $O/pytest_next.txt 2>&1; tail -3 $O/pytest_next.txt
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2; do
  for v in main next; do
    for cfg in c2:5 c2:10 c4:4; do
      c=${cfg%%:*}; n=${cfg##*:}
      OSRL_PIPE_DUAL=$v timeout 300 python bench.py --config $c --steps-per-graph $n $B > $O/b_${c}_${n}_${v}_$r.json 2> $O/b_${c}_${n}_${v}_$r.err
      echo "$c spg=$n $v r$r $(python -c "import json,sys; d=json.loads(open('$O/b_${c}_${n}_${v}_$r.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
    done
  done
done
(cd /tmp && OSRL_PIPE_DUAL=next rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps-per-graph 5 $B > $O/bench_profiled.json 2> $O/prof.err)
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline_graph.py $T 5 > $O/timeline_next_c2.txt 2>&1
rm -rf $O/prof
head -70 $O/timeline_next_c2.txt | cut -c1-110
