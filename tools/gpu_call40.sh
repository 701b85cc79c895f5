cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c40; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cdt.py -q > $O/t.log 2>&1; grep -E "passed|failed|Error|assert" $O/t.log | tail -8
timeout 300 python bench.py --config c5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-120
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cdt -o cdt -- python $GRAFT_REPO_ROOT/tools/prof_one.py cdt 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
S=$(find gpurun_out/prof_cdt -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/c41_cdt_kernel_stats.csv; rm -rf gpurun_out/prof_cdt
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c41_cdt_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:8]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
