cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c40; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cdt.py -q > $O/t.log 2>&1; grep -E "passed|failed|Error" $O/t.log | tail -8
timeout 300 python bench.py --config c5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-120
