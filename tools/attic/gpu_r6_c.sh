#!/bin/bash
# round 6: what the graph executor does with the pipelined graphs -- kernel-trace timelines of C3 / C2 at 4 steps per graph
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6c; rm -rf $O; mkdir -p $O
for cfg in c3 c2; do
  st=200; [ $cfg = c3 ] && st=80
  (cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof_$cfg -o bench -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --no-cold --steps $st --warmup 20 --steps-per-graph 4 > $O/bench_profiled_$cfg.json 2> $O/prof_$cfg.err)
  T=$(find $O/prof_$cfg -name "*kernel_trace.csv" | head -1)
  python tools/timeline_graph.py $T 4 > $O/timeline_4step_$cfg.txt 2>&1
  rm -rf $O/prof_$cfg
  cut -c1-120 $O/bench_profiled_$cfg.json; head -3 $O/timeline_4step_$cfg.txt; tail -2 $O/timeline_4step_$cfg.txt
done
timeout 600 python -m pytest "tests/test_gpu_train_step.py::test_random_shape_tuples_match_the_oracle" -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
# BC one-launch step: per-phase and per-layer stamps
OSRL_LIB=osrl_amd/lib/libosrl_stamps.so timeout 300 python tools/step_stamps.py 256 256 > $O/bc_phase.txt 2>&1; tail -30 $O/bc_phase.txt
