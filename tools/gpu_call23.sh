#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for lib in base prio; do
  if [ $lib = prio ]; then export OSRL_LIB=$PWD/osrl_amd/lib/libosrl_amd_prio.so; else unset OSRL_LIB; fi
  for t in 0 80; do for cap in 512 0; do
    [ $t = 80 ] && [ $cap = 512 ] && continue
    v=$(OSRL_OOD_TILE=$t OSRL_OOD_WG_CAP=$cap timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['value'])")
    echo "lib=$lib tile=$t cap=$cap steps/s=$v"
  done; done
  v=$(OSRL_ENC_TILE=80 timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['value'])")
  echo "lib=$lib enc80 only steps/s=$v"
done
