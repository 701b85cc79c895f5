#!/bin/bash
# round 6: pipelined CPQ steps with the two streams swapping roles from step to step -- bit-equality, A/B, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for cfg in c2 c4; do
  for rep in 1 2; do
  for sw in 0 1; do
  for spg in 2 4 10; do
    OSRL_LAB=1 OSRL_PIPE_SWAP=$sw timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 20 --steps-per-graph $spg > $O/b_${cfg}_${sw}_$spg.json 2>>$O/bench.err
    python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'swap', sys.argv[4], 'spg', sys.argv[3], d['value'], d['no_preroll']['value'])" $O/b_${cfg}_${sw}_$spg.json $cfg $spg $sw
  done; done
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 20 --steps-per-graph 1 > $O/b_${cfg}_1.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'one step per graph', d['value'], d['no_preroll']['value'])" $O/b_${cfg}_1.json $cfg
  done
done 2>&1 | tee $O/sweep.txt
(cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof_c2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c2 --no-cpu-baseline --no-extras --no-roofline --no-cold --steps 200 --warmup 20 --steps-per-graph 4 > $O/bench_profiled_c2.json 2> $O/prof_c2.err)
T=$(find $O/prof_c2 -name "*kernel_trace.csv" | head -1)
python tools/timeline_graph.py $T 4 > $O/timeline_4step_swap_c2.txt 2>&1
rm -rf $O/prof_c2
head -3 $O/timeline_4step_swap_c2.txt; tail -2 $O/timeline_4step_swap_c2.txt
