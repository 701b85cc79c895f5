#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_cdt -o cdt -- python $GRAFT_REPO_ROOT/tools/prof_one.py cdt 5 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
S=$(find gpurun_out/prof_cdt -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/c41_cdt_kernel_stats.csv; rm -rf gpurun_out/prof_cdt
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c41_cdt_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
