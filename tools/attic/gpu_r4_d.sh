#!/bin/bash
# round 4, step D: seeded backward launches -- bit-equality vs the loss launches, parity suite, A/B on the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4d; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_step.py -q -x -k "seeded" > $O/t0.log 2>&1; tail -15 $O/t0.log
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_dp_sim.py -q -x -k "cpq" > $O/t1.log 2>&1; tail -5 $O/t1.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
for rep in 1 2 3; do for v in 0 1; do echo "OSRL_SEEDS=$v"; env OSRL_SEEDS=$v $B 2>>$O/bench.err | cut -c1-70; done; done
for v in 0 1; do echo "c4 OSRL_SEEDS=$v"; env OSRL_SEEDS=$v $B --config c4 2>>$O/bench.err | cut -c1-70; done
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
rm -rf $O/prof
cat $O/timeline.txt
