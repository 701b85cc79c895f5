cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c37; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train_step.py -q -k "launch_plan" -x > $O/t.log 2>&1; tail -5 $O/t.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
echo "PLAN=1"; $B 2>>$O/bench.err | cut -c1-60
for cap in 512 256 0; do echo "PLAN=3 half cap=$cap"; OSRL_CPQ_PLAN=3 OSRL_OOD_WG_CAP=$cap $B 2>>$O/bench.err | cut -c1-60; done
echo "PLAN=3 nomask"; OSRL_CPQ_PLAN=3 OSRL_CPQ_MASKS=none $B 2>>$O/bench.err | cut -c1-60
echo "PLAN=3 swap"; OSRL_CPQ_PLAN=3 OSRL_CPQ_MASKS=swap $B 2>>$O/bench.err | cut -c1-60
echo "PLAN=3 half eager"; OSRL_CPQ_PLAN=3 OSRL_PLAN_SEGMENTS=0 $B 2>>$O/bench.err | cut -c1-60
cd /tmp && OSRL_CPQ_PLAN=3 rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline_p3.txt 2>&1
rm -rf $O/prof
cat $O/timeline_p3.txt
