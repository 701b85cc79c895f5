#!/bin/bash
# round-6 THIRD-SESSION evidence with the final library (-> profiles/r6c_*: plan.ood_rows as the rule at C2 / C4, the two new edges of the no-join graphs): smoke, two runs of the driver's command (full line: roofline,
# cpu_baseline, cost_return_gap, other configs), a 300-step run, the rocprofv3 kernel-trace stats / per-(kernel, grid)
# summary / multi-step-graph timeline of the C2 bench (5 steps per graph) and of C3 (10 per graph), C4 / C5 per-kernel
# stats, forced data parallelism on one rank (C2, C4)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6cfinal; rm -rf $O; mkdir -p $O
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$i.json 2>>$O/bench.err; python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd', d['value'], d['no_preroll'], d['config']['steps_per_graph'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['isolated_frac'], {k:(v.get('steps_per_s'), v.get('steps_per_graph')) for k,v in d['other_configs'].items()}, d.get('config',{}).get('cost_return_gap_vs_ref'))" $O/bench_driver_cmd_$i.json; done
timeout 600 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_300.json 2>>$O/bench.err; cut -c1-90 $O/bench_300.json
for cfg in c2 c3; do
  st=400; spg=20; [ $cfg = c3 ] && st=100 && spg=10
  (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$cfg -o bench -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline --no-extras --steps $st --warmup 20 > $O/bench_profiled_$cfg.json 2> $O/prof_$cfg.err)
  cp $(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_$cfg.csv
  T=$(find $O/prof_$cfg -name "*kernel_trace.csv" | head -1)
  python tools/timeline_graph.py $T $spg > $O/timeline_${spg}step_$cfg.txt 2>&1
  python tools/trace_summary.py $T > $O/trace_summary_$cfg.txt 2>&1
  rm -rf $O/prof_$cfg
  cut -c1-100 $O/bench_profiled_$cfg.json; head -3 $O/timeline_${spg}step_$cfg.txt; tail -1 $O/timeline_${spg}step_$cfg.txt
done
for cfg in c4 c5; do
  st=200; [ $cfg = c5 ] && st=10
  (cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$cfg -o bench -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps $st --warmup 3 > $O/bench_profiled_$cfg.json 2> $O/prof_$cfg.err)
  cp $(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_$cfg.csv
  python tools/trace_summary.py $(find $O/prof_$cfg -name "*kernel_trace.csv" | head -1) > $O/trace_summary_$cfg.txt 2>&1
  rm -rf $O/prof_$cfg
  cut -c1-100 $O/bench_profiled_$cfg.json
done
OSRL_FORCE_DP=1 timeout 600 python bench.py --steps 300 --warmup 20 > $O/bench_c2_forced_dp.json 2>>$O/bench.err; cut -c1-80 $O/bench_c2_forced_dp.json
OSRL_FORCE_DP=1 timeout 600 python bench.py --config c4 --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_c4_forced_dp.json 2>>$O/bench.err; cut -c1-80 $O/bench_c4_forced_dp.json
timeout 300 python bench.py --config c4 --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_c4.json 2>>$O/bench.err; cut -c1-80 $O/bench_c4.json
tail -3 $O/bench.err
# un-profiled kernel start stamps of the shipped C2 / C4 graphs (csrc/trace.h lab build)
for cfg in c2 c4; do
  OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_trace.so timeout 300 python tools/trace_steps.py $cfg 0 40 > $O/trace_unprofiled_$cfg.txt 2>> $O/bench.err
  head -2 $O/trace_unprofiled_$cfg.txt
done
# the rocprofv3 kernel stats of the C2 bench on the ONE-step graph (the graph the roofline's in-graph stamps are taken on)
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_c2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps-per-graph 1 --no-cpu-baseline --no-extras --steps 400 --warmup 20 > $O/bench_profiled_c2_1step.json 2> $O/prof_c2.err)
cp $(find $O/prof_c2 -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats_c2_1step.csv
python tools/timeline_graph.py $(find $O/prof_c2 -name "*kernel_trace.csv" | head -1) 1 > $O/timeline_1step_c2.txt 2>&1
rm -rf $O/prof_c2
grep nb8 $O/bench_kernel_stats_c2_1step.csv | cut -c1-120
