#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cdt.py tests/test_gpu_dp_sim.py tests/test_gpu_data_eval.py -m gpu -q --timeout=600 -k "cdt or layernorm" > gpurun_out/c42_pytest.log 2>&1; grep -E "passed|failed|Error" gpurun_out/c42_pytest.log | head -5
timeout 200 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('c5 steps/s', d['value'], d['ms_per_step'])"
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cdt -o cdt -- python $GRAFT_REPO_ROOT/tools/prof_one.py cdt 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
S=$(find gpurun_out/prof_cdt -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/c42_cdt_kernel_stats.csv; find gpurun_out/prof_cdt -name "*trace.csv" -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c42_cdt_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
