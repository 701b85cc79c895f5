#!/bin/bash
# round 6 (second session): the critic group's dW tile now that its launch (side branch, 651-676 us) no longer overlaps the
# N*B-row encoder launch (704 us on): 32 x 32 x 2 splits (shipped) against larger tiles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6dwt; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2; do
  for v in "2 2" "0 0" "4 1" "4 2" "3 2" "3 3"; do
    set -- $v
    if [ $1 = 0 ]; then E="OSRL_DW_T_CRITIC=0"; else E="OSRL_DW_T_CRITIC=$1 OSRL_DW_S_CRITIC=$2"; fi
    env $E timeout 300 python bench.py --config c2 $B > $O/b.json 2> $O/b.err
    echo "c2 critic dW tile=$1 splits=$2 r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
