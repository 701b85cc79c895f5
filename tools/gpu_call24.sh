#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for e in 0 80; do for i in 1 2; do
  v=$(OSRL_ENC_TILE=$e timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['value'])")
  echo "enc_tile=$e steps/s=$v"
done; done
