#!/bin/bash
# round-4 evidence with the final library: full GPU suite + smoke, two runs of the driver's command, the rocprofv3
# kernel-trace stats / per-(kernel, grid) summary / timeline of the bench, kernarg placement A/B, the scaling script at N = 1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$i.json 2>>$O/bench.err; python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd', d['value'], d['no_preroll'], d['roofline']['frac'], d['roofline']['isolated_frac'], {k:v.get('steps_per_s') for k,v in d['other_configs'].items()})" $O/bench_driver_cmd_$i.json; done
timeout 600 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_300.json 2>>$O/bench.err; cut -c1-90 $O/bench_300.json
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
rm -rf $O/prof
bash tools/gpu_r4_kernarg.sh > /dev/null 2>&1; cp gpurun_out/r4k/kernarg_ab.txt $O/kernarg_ab.txt
NS="1" bash tools/scale_run.sh $O/scale 2>&1 | tail -6
