#!/bin/bash
# round 6 (second session): s_setprio in the all-CU VAE launches (csrc/vae_ns.hip OSRL_VAE_NS_PRIO) -- alt libraries through OSRL_LIB
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6prio; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
for r in 1 2 3; do
  for v in base _p3 _p2; do
    for cfg in c2 c4; do
      if [ $v = base ]; then E="X=1"; else E="OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_alt$v.so"; fi
      env $E timeout 300 python bench.py --config $cfg $B > $O/b.json 2> $O/b.err
      echo "$cfg vae_ns prio $v r$r $(python -c "import json,sys; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
    done
  done
done
