cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c34; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | tail -3
timeout 300 python bench.py --no-cpu-baseline 2>>$O/bench.err > $O/bench.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/c34/bench.json")); print(d["value"], d["api_path"], d["other_configs"])
PY
