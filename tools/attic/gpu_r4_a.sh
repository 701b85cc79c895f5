#!/bin/bash
# round 4, step A: the fused dW + Adam launch -- its bit-equality test, the train-step parity suite, A/B on the bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4a; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "dw_tiles_adam or adam" > $O/t0.log 2>&1; tail -15 $O/t0.log
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_bc_one_launch.py -q -x > $O/t1.log 2>&1; tail -5 $O/t1.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
for rep in 1 2 3; do for v in 0 1; do echo "OSRL_FUSE_DW_ADAM=$v"; env OSRL_FUSE_DW_ADAM=$v $B 2>>$O/bench.err | cut -c1-70; done; done
for v in 0 1; do echo "c4 OSRL_FUSE_DW_ADAM=$v"; env OSRL_FUSE_DW_ADAM=$v $B --config c4 2>>$O/bench.err | cut -c1-70; done
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
rm -rf $O/prof
cat $O/timeline.txt
