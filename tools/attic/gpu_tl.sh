#!/bin/bash
# headline bench (x2) + a step timeline from a rocprofv3 kernel trace:  gpurun -- 'bash tools/gpu_tl.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/tl; rm -rf $O; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.readline()); r=d.get("roofline",{}); print(d["value"], d["ms_per_step"], r.get("frac"), r.get("in_step_us"), r.get("kernels"))'
for rep in 1 2; do timeout 300 python bench.py --no-extras --no-cpu-baseline 2>>$O/bench.err | python -c "$P"; done
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
cp $T $O/kernel_trace.csv; rm -rf $O/prof
cat $O/timeline.txt
