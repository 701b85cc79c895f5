#!/bin/bash
# round-6 second session, evidence re-taken with the final bench.py: two runs of the driver's command, a 300-step run, and the
# rocprofv3 kernel stats of the C2 bench on the ONE-step graph (the graph the roofline's in-graph stamps are taken on)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6bfinal2; rm -rf $O; mkdir -p $O
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$i.json 2>>$O/bench.err; python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd', d['value'], d['no_preroll'], d['config']['steps_per_graph'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['isolated_frac'], {k:(v.get('steps_per_s'), v.get('steps_per_graph')) for k,v in d['other_configs'].items()})" $O/bench_driver_cmd_$i.json; done
timeout 600 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_300.json 2>>$O/bench.err; cut -c1-90 $O/bench_300.json
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_c2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps-per-graph 1 --no-cpu-baseline --no-extras --steps 400 --warmup 20 > $O/bench_profiled_c2_1step.json 2> $O/prof_c2.err)
cp $(find $O/prof_c2 -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_c2_1step.csv
python tools/timeline_graph.py $(find $O/prof_c2 -name "*kernel_trace.csv" | head -1) 1 > $O/timeline_1step_c2.txt 2>&1
rm -rf $O/prof_c2
grep nb8 $O/bench_kernel_stats_c2_1step.csv | cut -c1-120
tail -3 $O/bench.err
