#!/bin/bash
# round 6: pipelined CPQ steps, placement of the dual step / the next prologue (lab knobs), C2 and C4; PMC re-take
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6e; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() {  # cfg spg label env...
  cfg=$1; spg=$2; lab=$3; shift 3
  env OSRL_LAB=1 "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 20 --steps-per-graph $spg > $O/b.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'spg', sys.argv[3], sys.argv[4], d['value'], d['no_preroll']['value'])" $O/b.json $cfg $spg "$lab"
}
for rep in 1 2; do
for cfg in c2 c4; do
  run $cfg 1 one-step X=0
  for spg in 2 4; do
    run $cfg $spg default X=0
    run $cfg $spg dual-side OSRL_PIPE_DUAL=side
    run $cfg $spg prologue-early OSRL_PIPE_PROLOGUE=early
    run $cfg $spg dual-side+early OSRL_PIPE_DUAL=side OSRL_PIPE_PROLOGUE=early
    run $cfg $spg prologue-main OSRL_PIPE_PROLOGUE=main
  done
done; done 2>&1 | tee $O/sweep.txt
tail -3 $O/bench.err
