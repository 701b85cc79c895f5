#!/bin/bash
# rocprofv3 kernel trace of bench.py (the driver's command line), summaries -> gpurun_out/prof_step/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_step
rm -rf $OUT; mkdir -p $OUT
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-extras ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
T=$(find $OUT -name "*kernel_trace.csv" | head -1); S=$(find $OUT -name "*kernel_stats.csv" | head -1)
python tools/timeline.py $T > $OUT/timeline.txt 2>&1
python tools/trace_summary.py $T > $OUT/trace_summary.txt 2>&1
cp $S $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/timeline.txt | head -60
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['roofline']['isolated_us'], d['roofline']['in_step_us'], d['roofline']['kernels'])"
