#!/bin/bash
# A/B of environment switches on the headline bench:  gpurun -- 'bash tools/gpu_env_ab.sh "A=1" "B=2 C=3" ...'
# (each argument is one variant's environment; the empty string "" = the defaults).  REPS=1|2 (default 2), CFG=c2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --config ${CFG:-c2}"
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do echo -n "[$v] "; env $v $B 2>/dev/null | python -c "$P"; done
done
