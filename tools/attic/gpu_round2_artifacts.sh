#!/bin/bash
# Round-2 evidence run: full GPU suite, the driver's bench line, rocprofv3 kernel trace + stats of the same command,
# PMC traffic of the dominant kernel, evaluation throughput, all-config throughputs.  Outputs -> gpurun_out/r2/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2; rm -rf $O; mkdir -p $O
rm -f gpurun_out/parity_margins.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
cp gpurun_out/parity_margins.txt $O/ 2>/dev/null
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_driver_cmd.json 2>> $O/bench.err
OSRL_FORCE_DP=1 timeout 300 python bench.py --gpus 1 --config c4 --no-cpu-baseline > $O/bench_c4_forced_dp.json 2>> $O/bench.err
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1); S=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
python tools/trace_summary.py $T > $O/trace_summary.txt 2>&1
cp $S $O/bench_kernel_stats.csv
rm -rf $O/prof
timeout 300 python tests/bench_eval.py > $O/eval_bench.json 2> $O/eval.err
timeout 120 python tools/kbench.py --glue > $O/kbench_chain.txt 2>&1
for a in "20480 80 1 400 8" "20480 80 2 256 1"; do echo "# tools/mlp_phase.bin $a"; timeout 60 ./tools/mlp_phase.bin $a; done > $O/mlp_phase_nb.txt 2>&1
tail -4 $O/pytest.log; head -c 1500 $O/bench.json; echo; head -3 $O/bench_kernel_stats.csv | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
