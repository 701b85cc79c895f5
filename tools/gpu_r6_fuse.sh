#!/bin/bash
# round 6 (third session): fused dW + Adam launches forced on the split dW plans (operator switch OSRL_FUSE_DW_ADAM=1) under the final plan
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6fuse; rm -rf $O; mkdir -p $O
OSRL_FUSE_DW_ADAM=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ood_rows.py tests/test_gpu_bench_path.py -x -q -k "c2 or c4" > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
export OSRL_LAB=1
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in c2:auto c2:1 c4:auto c4:1; do
    IFS=: read cfg f <<< "$v"
    OSRL_FUSE_DW_ADAM=$f timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
    echo "$cfg fuse_dw_adam=$f r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
tail -n 2 $O/b.err
