cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c31; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "prologue or quantile or ood_stat or randn" > $O/t.log 2>&1; tail -3 $O/t.log
timeout 900 python -m pytest tests/test_gpu_train_step.py -q -k cpq > $O/t2.log 2>&1; tail -3 $O/t2.log
for f in 0 1; do OSRL_COST_FWD_SIDE=$f timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-60; done
for f in 0 1; do OSRL_FUSED_BEGIN=$f timeout 300 python bench.py --config c1 --steps 2000 --warmup 100 --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-60; done
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
rm -rf $O/prof
cat $O/timeline.txt
