#!/bin/bash
# round-3 evidence on ONE lease: the driver's command (full line: roofline, lease facts, api_path, other configs, cpu baseline),
# then rocprofv3 kernel stats / trace summary / timeline of the same workload.   gpurun -- 'bash tools/gpu_r3_final.sh TAG'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=${1:-lease}
O=$GRAFT_REPO_ROOT/gpurun_out/r3_$TAG; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_driver_cmd.json')); print('driver cmd:', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'in-step', d['roofline']['in_step_us'], 'exec', d['step_frac_executed'], 'api', d['api_path']['steps_per_s'], {k: v.get('steps_per_s') for k, v in d['other_configs'].items()}, d['lease']['kernargs_in'])"
if [ "$2" = "prof" ]; then
  timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_long.json 2>> $O/bench.err
  python -c "import json; d=json.load(open('$O/bench_long.json')); print('300 steps:', d['value'], d['ms_per_step'])"
  (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 200 > $O/bench_profiled.json 2> $O/prof.err)
  T=$(find $O/prof -name "*kernel_trace.csv" | head -1); K=$(find $O/prof -name "*kernel_stats.csv" | head -1)
  python tools/timeline.py $T > $O/timeline.txt 2>&1
  python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
  cp $K $O/kernel_stats.csv; rm -rf $O/prof
  head -8 $O/kernel_stats.csv | cut -c1-160
fi
