#!/bin/bash
# round 6: the driver's command (K = 20, W = 5) at 1 / 2 / 4 / 5 / 10 steps per graph, alternating, 5 rounds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6g; rm -rf $O; mkdir -p $O
for rep in 1 2 3 4 5; do
  for spg in 1 2 4 5 10; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline --steps-per-graph $spg > $O/b.json 2>>$O/bench.err
    python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('K=20 spg', sys.argv[2], d['value'], d['no_preroll']['value'])" $O/b.json $spg
  done
done 2>&1 | tee $O/sweep.txt
python - <<'PY'
import re, statistics, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6g"
acc = {}
for l in open(O + "/sweep.txt"):
    m = re.match(r"K=20 spg (\d+) ([\d.]+) ([\d.]+)", l)
    if m: acc.setdefault(int(m.group(1)), []).append(float(m.group(2)))
for k, v in sorted(acc.items()): print("spg", k, "median", statistics.median(v), "min", min(v), "max", max(v))
PY
