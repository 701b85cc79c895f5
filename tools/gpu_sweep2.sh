#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -f gpurun_out/sweep2.txt
for ecap in 0 1024 768 512 384; do for ocap in 512 768 384; do
  v=$(OSRL_ENC_WG_CAP=$ecap OSRL_OOD_WG_CAP=$ocap timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; print(json.load(sys.stdin)['value'])")
  echo "enc_cap=$ecap ood_cap=$ocap steps/s=$v" >> gpurun_out/sweep2.txt
done; done
cat gpurun_out/sweep2.txt
