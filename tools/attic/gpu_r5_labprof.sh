#!/bin/bash
# round 5: rocprofv3 kernel trace of a standalone lab binary (per-kernel durations without torch)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export OSRL_LAB=1  # lab switches (OSRL_*) are read only under this (engine/plan.py)
O=$GRAFT_REPO_ROOT/gpurun_out/r5lab; mkdir -p $O
b=$1
(cd /tmp && HIP_FORCE_DEV_KERNARG=${KERNARG:-1} rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$b -o p -- $GRAFT_REPO_ROOT/tools/_lab/$b > $O/${b}_prof.log 2>&1)
S=$(find $O/prof_$b -name "*kernel_stats.csv" | head -1)
cp "$S" $O/${b}_kernel_stats.csv 2>/dev/null
T=$(find $O/prof_$b -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per (kernel, grid) durations
d = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"][:60]
    g = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
    d[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(d.items()):
    v.sort()
    print(f"{k:62s} grid {g:>8s} n {len(v):4d}  median {v[len(v)//2]:8.2f} us  min {v[0]:8.2f}")
PY
rm -rf $O/prof_$b
