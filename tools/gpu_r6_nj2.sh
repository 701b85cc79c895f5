#!/bin/bash
# round 6 (second session): why the no-join graph's kernel timeline (430 us/step) and the bench's clock (449) disagree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6nj2; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --warmup 20"
for v in main next; do
  for k in 300 1500; do
    OSRL_PIPE_DUAL=$v timeout 300 python bench.py --config c2 --steps-per-graph 5 --steps $k $B > $O/b_${v}_$k.json 2> $O/b_${v}_$k.err
    echo "c2 spg=5 $v K=$k $(python -c "import json,sys; d=json.loads(open('$O/b_${v}_$k.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))")" | tee -a $O/ab.txt
  done
  (cd /tmp && OSRL_PIPE_DUAL=$v rocprofv3 --kernel-trace -f csv -d $O/prof_$v -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps-per-graph 5 --steps 300 $B > $O/bench_profiled_$v.json 2> $O/prof_$v.err)
  T=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1)
  python tools/timeline_graph.py $T 5 > $O/timeline_${v}_c2.txt 2>&1
  python - $T > $O/replay_durs_$v.txt <<'P'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id')) for r in rows)
al = [k for k in ks if 'cpq_alpha_step' in k[2]]
# every 5th alpha step closes a graph: print the interval between consecutive alpha-step ENDS, per step
prev = None
out = []
for k in al:
    if prev is not None:
        out.append((k[1] - prev) / 1e3)
    prev = k[1]
print(' '.join('%.0f' % x for x in out))
P
  rm -rf $O/prof_$v
  head -3 $O/timeline_${v}_c2.txt
done
