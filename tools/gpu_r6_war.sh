#!/bin/bash
# round 6 (third session): C4's no-join graph with the VAE's optimizer step ordered behind (an edge of its own: WAR against
# the previous step's N*B-row encoder launch)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6war; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ood_rows.py tests/test_gpu_pipeline.py -x -q > $O/pytest.txt 2>&1; tail -n 3 $O/pytest.txt
export OSRL_LAB=1
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3 4; do
  for v in c4:actor c4:vae c4:0; do
    IFS=: read cfg w <<< "$v"
    OSRL_VAE_WAR_EDGE=$w timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
    echo "$cfg war_edge=$w r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
done
tail -n 2 $O/b.err
