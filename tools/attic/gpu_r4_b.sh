#!/bin/bash
# round 4, step B: the 8-wave N*B-row encoder kernel (<= 152 registers) -- parity, isolated time, A/B in the step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4b; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "big_rows" > $O/t0.log 2>&1; tail -5 $O/t0.log
timeout 900 python -m pytest tests/test_gpu_train_step.py -q -x -k "cpq or full" > $O/t1.log 2>&1; tail -3 $O/t1.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline"
for rep in 1 2 3; do for v in 4 8; do echo "OSRL_NB_WAVES=$v"; env OSRL_NB_WAVES=$v $B 2>>$O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('roofline', {})
print(d['value'], 'isolated_us', r.get('isolated_us'), 'in_step_us', r.get('in_step_us'), 'sites', r.get('in_step_sites_us'))"; done; done
for v in 4 8; do echo "c4 OSRL_NB_WAVES=$v"; env OSRL_NB_WAVES=$v $B --no-roofline --config c4 2>>$O/bench.err | cut -c1-70; done
for v in 4 8; do echo "c3 OSRL_NB_WAVES=$v"; env OSRL_NB_WAVES=$v $B --no-roofline --config c3 2>>$O/bench.err | cut -c1-70; done
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
rm -rf $O/prof
cat $O/timeline.txt
