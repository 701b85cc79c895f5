#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/sweep_cap.txt
for rep in 1 2; do
for cap in 512 384 256 192 128 96 64; do
  v=$(OSRL_OOD_WG_CAP=$cap timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['value'])")
  echo "cap=$cap rep=$rep steps/s=$v" >> gpurun_out/sweep_cap.txt
done; done
cat gpurun_out/sweep_cap.txt
