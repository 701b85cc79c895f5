#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for i in 1 2; do timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('steps/s', d['value'], d['last_stats'])"; done
bash tools/gpu_prof_step.sh > gpurun_out/c18_prof.txt 2>&1; head -45 gpurun_out/c18_prof.txt | cut -c1-110
