cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c43; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/t.log 2>&1; grep -E "passed|failed|Error|error|assert" $O/t.log | tail -12
timeout 300 python bench.py --config c5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-100
