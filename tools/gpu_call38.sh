cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c38; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train_step.py -q -k "cpq" > $O/t.log 2>&1; tail -3 $O/t.log
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
for rep in 1 2; do for v in 0 1; do echo "VAE_DW_SIDE=$v"; OSRL_VAE_DW_SIDE=$v $B 2>>$O/bench.err | cut -c1-60; done; done
for v in 0 1; do echo "c4 VAE_DW_SIDE=$v"; OSRL_VAE_DW_SIDE=$v $B --config c4 2>>$O/bench.err | cut -c1-60; done
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
rm -rf $O/prof
cat $O/timeline.txt
