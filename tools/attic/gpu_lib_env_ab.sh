#!/bin/bash
# A/B of library builds x environments on the headline bench:
#   gpurun -- 'bash tools/gpu_lib_env_ab.sh "LIBS" "ENV A" "ENV B" ...'   LIBS = space-separated .so paths ("-" = the tree's)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
LIBS=$1; shift
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --config ${CFG:-c2}"
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])'
for rep in $(seq 1 ${REPS:-2}); do
  for l in $LIBS; do
    for v in "$@"; do
      if [ "$l" = "-" ]; then echo -n "[tree | $v] "; env $v $B 2>/dev/null | python -c "$P";
      else echo -n "[$l | $v] "; env $v OSRL_LIB=$GRAFT_REPO_ROOT/$l $B 2>/dev/null | python -c "$P"; fi
    done
  done
done
