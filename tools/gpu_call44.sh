cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c44; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_dp_sim.py -q > $O/t.log 2>&1; grep -E "passed|failed|Error|error|assert" $O/t.log | tail -6
timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-60
timeout 300 python bench.py --config c3 --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-60
