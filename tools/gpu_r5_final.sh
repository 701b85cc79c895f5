#!/bin/bash
# round-5 evidence with the final library (-> profiles/r5_*): smoke, two runs of the driver's command (full line: roofline,
# cpu_baseline, cost_return_gap, other configs), a 300-step run, the rocprofv3 kernel-trace stats / per-(kernel, grid)
# summary / timeline of the C2 bench and of C4 (the plan with the all-CU VAE launches), forced data parallelism on one rank
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5final; rm -rf $O; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$i.json 2>>$O/bench.err; python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd', d['value'], d['no_preroll'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['isolated_frac'], {k:v.get('steps_per_s') for k,v in d['other_configs'].items()}, d.get('config',{}).get('cost_return_gap_vs_ref'))" $O/bench_driver_cmd_$i.json; done
timeout 600 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_300.json 2>>$O/bench.err; cut -c1-90 $O/bench_300.json
for cfg in c2 c4; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$cfg -o bench -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline --no-extras --steps 200 > $O/bench_profiled_$cfg.json 2> $O/prof_$cfg.err)
  cp $(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_$cfg.csv
  T=$(find $O/prof_$cfg -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py $T > $O/timeline_$cfg.txt 2>&1
  python tools/trace_summary.py $T > $O/trace_summary_$cfg.txt 2>&1
  rm -rf $O/prof_$cfg
done
OSRL_FORCE_DP=1 timeout 600 python bench.py --steps 300 --warmup 20 > $O/bench_c2_forced_dp.json 2>>$O/bench.err; cut -c1-80 $O/bench_c2_forced_dp.json
# C3 (BCQ-Lag) timeline and C5 (CDT) per-kernel stats of the final library
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_c3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c3 --no-cpu-baseline --no-extras --no-roofline --steps 100 --warmup 10 > $O/bench_profiled_c3.json 2> $O/prof_c3.err)
T=$(find $O/prof_c3 -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline_c3.txt 2>&1
python tools/trace_summary.py $T > $O/trace_summary_c3.txt 2>&1
rm -rf $O/prof_c3
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_c5 -o cdt -- python $GRAFT_REPO_ROOT/bench.py --config c5 --no-cpu-baseline --no-extras --no-roofline --steps 10 --warmup 3 > $O/bench_profiled_c5.json 2> $O/prof_c5.err)
cp $(find $O/prof_c5 -name "*kernel_stats.csv" | head -1) $O/cdt_kernel_stats.csv
python tools/trace_summary.py $(find $O/prof_c5 -name "*kernel_trace.csv" | head -1) > $O/cdt_trace_summary.txt 2>&1
rm -rf $O/prof_c5
cut -c1-100 $O/bench_profiled_c3.json $O/bench_profiled_c5.json
head -12 $O/timeline_c3.txt
