#!/bin/bash
# round 6 (third session): the VAE Adam's edge to the main chain in the no-join graphs (C2), and the VAE Adam on the side branch at C4
# now that plan.ood_rows took the N*B-row cost-critic launch out of that branch's first half
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6vaeedge; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ood_rows.py tests/test_gpu_pipeline.py -x -q > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
export OSRL_LAB=1
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in c2:auto:0 c2:auto:actor c2:auto:next c4:0:actor c4:1:actor c4:1:next; do
    IFS=: read cfg s e <<< "$v"
    OSRL_VAE_ADAM_SIDE=$s OSRL_VAE_ADAM_EDGE=$e timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
    echo "$cfg vae_adam_side=$s edge=$e r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
tail -n 2 $O/b.err
