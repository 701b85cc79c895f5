#!/bin/bash
# round 6 (third session): plan.ood_rows in the data-parallel step (global quantile, per-rank row lists): DP tests, forced-DP A/B on one rank
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6dprows; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dp_sim.py tests/test_gpu_ipc_dp.py -x -q > $O/pytest.txt 2>&1; tail -n 5 $O/pytest.txt
export OSRL_LAB=1
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in c2:1 c2:0 c4:1 c4:0; do
    IFS=: read cfg d <<< "$v"
    OSRL_FORCE_DP=1 OSRL_OOD_ROWS_DP=$d timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
    echo "$cfg forced-DP ood_rows_dp=$d r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
tail -n 2 $O/b.err
