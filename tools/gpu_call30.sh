#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for s in 0 1; do for t in 0 80; do
  v=$(OSRL_BCQ_SERIAL=$s OSRL_BCQ_TILE=$t timeout 200 python bench.py --config c3 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'])")
  echo "serial=$s tile=$t steps/s=$v"
done; done
