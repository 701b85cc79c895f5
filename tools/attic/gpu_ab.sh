#!/bin/bash
# A/B of one environment switch on the headline bench, plus the CPQ parity tests and a step timeline:
#   gpurun -- 'bash tools/gpu_ab.sh OSRL_CPQ_MERGED_Q 0 1'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
VAR=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_dp_sim.py tests/test_gpu_kernels.py -q -k "cpq or quantile or mlp" > $O/t.log 2>&1; grep -E "passed|failed|Error|assert" $O/t.log | tail -6
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
for rep in 1 2; do for v in "$@"; do echo "$VAR=$v"; env $VAR=$v $B 2>>$O/bench.err | cut -c1-60; done; done
for v in "$@"; do echo "c4 $VAR=$v"; env $VAR=$v $B --config c4 2>>$O/bench.err | cut -c1-60; done
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
rm -rf $O/prof
cat $O/timeline.txt
