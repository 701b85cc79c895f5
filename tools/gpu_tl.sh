#!/bin/bash
# step timeline of the headline bench under the environment given as arguments: gpurun -- 'bash tools/gpu_tl.sh VAR=val ...'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tl; mkdir -p $O
cd /tmp && env "$@" rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
rm -rf $O/prof
cut -c1-60 $O/bench_profiled.json; cat $O/timeline.txt
