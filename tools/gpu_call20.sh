#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for of in 1 0; do for cap in 512 0; do
  v=$(OSRL_OOD_FIRST=$of OSRL_OOD_WG_CAP=$cap timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['value'])")
  echo "ood_first=$of ood_cap=$cap steps/s=$v"
done; done
