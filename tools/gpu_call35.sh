cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c35; mkdir -p $O; export TMPDIR=/tmp
OSRL_CPQ_PLAN=2 timeout 600 python -m pytest tests/test_gpu_train_step.py -q -k cpq > $O/t2.log 2>&1; tail -3 $O/t2.log
OSRL_CPQ_PLAN=2 OSRL_P2_LOOP_EARLY=1 timeout 600 python -m pytest tests/test_gpu_train_step.py -q -k "cpq and parallel" > $O/t3.log 2>&1; tail -3 $O/t3.log
for rep in 1 2; do
for v in "1 0" "2 0" "2 1"; do set -- $v; echo "PLAN=$1 LOOP_EARLY=$2"; OSRL_CPQ_PLAN=$1 OSRL_P2_LOOP_EARLY=$2 timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-60; done; done
cd /tmp && OSRL_CPQ_PLAN=2 rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline_p2.txt 2>&1
rm -rf $O/prof
cat $O/timeline_p2.txt
