#!/bin/bash
# round 6 (third session): plan.ood_rows as C4's rule (late side start): tests, then enc-share A/B at C4 and C2, driver command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6oodrows5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ood_rows.py tests/test_gpu_pipeline.py tests/test_gpu_bench_path.py -x -q > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
export OSRL_LAB=1
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in c4:0:1 c4:auto:0 c4:auto:1 c2:auto:1 c2:1:0 c2:1:1; do
    IFS=: read cfg o s <<< "$v"
    OSRL_OOD_ROWS=$o OSRL_OOD_ROWS_ENC_SHARE=$s timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
    echo "$cfg ood_rows=$o enc_share=$s r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
tail -n 2 $O/b.err
