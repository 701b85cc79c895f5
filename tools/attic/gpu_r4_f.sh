#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4f; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_dp_sim.py -q -x -k "cpq" > $O/t1.log 2>&1; tail -5 $O/t1.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; E=(); A=(); for x in "$@"; do case "$x" in *=*) E+=("$x");; *) A+=("$x");; esac; done; env "${E[@]}" $B "${A[@]}" 2>>$O/bench.err | cut -c1-60; }
for rep in 1 2; do
run OSRL_HEAD_TAILS=1
run OSRL_HEAD_TAILS=0
done
run OSRL_HEAD_TAILS=1 --config c4
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
rm -rf $O/prof
cat $O/timeline.txt
