#!/bin/bash
# round 6: the 13..16-block N*B-row form held to 256 registers per lane (205, accumulators in VGPRs; 285 = 205 + 80 AGPRs
# since the fused head) -- A/B against the previous build at C2 / C4, K = 200, three rounds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6q; rm -rf $O; mkdir -p $O
run() {  # cfg label env...
  cfg=$1; lab=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 20 > $O/b.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], sys.argv[3], d['value'], d['no_preroll']['value'], d['ms_per_step'])" $O/b.json $cfg "$lab"
}
for rep in 1 2 3; do
  for cfg in c2 c4; do
    run $cfg 205-registers X=0
    run $cfg 285-registers OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_prev.so
  done
done 2>&1 | tee $O/ab.txt
