#!/bin/bash
# round 6 (third session): with the VAE Adam's edge in place -- C2 with the VAE Adam back on the main chain, prologue placements under plan.ood_rows
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6vaeedge2; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in c2:auto:head c2:0:head c2:auto:critic c2:0:critic c4:0:critic c4:0:head; do
    IFS=: read cfg s p <<< "$v"
    OSRL_VAE_ADAM_SIDE=$s OSRL_PIPE_PROLOGUE=$p timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
    echo "$cfg vae_adam_side=$s prologue=$p r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
tail -n 2 $O/b.err
