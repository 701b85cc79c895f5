#!/bin/bash
# round 6: the driver's full command eight times with graph priming (is the first timed region's stall gone?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6p; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd', d['value'], d['no_preroll'], d['config']['steps_per_graph'], d['roofline']['frac'])" $O/b.json
done 2>&1 | tee $O/summary.txt
