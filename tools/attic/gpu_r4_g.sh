#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4g; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bc_one_launch.py tests/test_gpu_train_step.py -q -x -k "flagged or full_size" > $O/t0.log 2>&1; tail -4 $O/t0.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err ) 2>&1 | grep real
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4g/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("value", d["value"], "no_preroll", d["no_preroll"], "step_frac", d["step_frac"], d["step_frac_executed"])
print("roofline", {k: d["roofline"].get(k) for k in ("frac", "frac_is", "isolated_frac", "in_step_us", "isolated_us")})
print("other", {k: (v.get("steps_per_s"), (v.get("cpu_baseline") or {}).get("value")) for k, v in d["other_configs"].items()})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "api", d["api_path"]["steps_per_s"])
PY
OSRL_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_force_dp.json 2>> $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4g/bench_force_dp.json").read().strip().splitlines()[-1])
print("force_dp value", d["value"], "graph", d["config"]["graph"], "collectives", d["collectives_in_step"])
PY
bash tools/gpu_pmc_mfma.sh 2>&1 | tail -60
