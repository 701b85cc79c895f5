#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for f in none fwd both; do
OSRL_CDT_FUSE=$f timeout 200 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('fuse=$f c5 steps/s', d['value'], d['ms_per_step'], d['last_stats']['all_loss'])"
done
