#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in base pin1; do
  if [ $v = pin1 ]; then export OSRL_LIB=$PWD/osrl_amd/lib/libosrl_amd_pin1.so; else unset OSRL_LIB; fi
  timeout 300 python tools/kbench.py > gpurun_out/c5_kbench_$v.txt 2>&1
  timeout 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c5_bench_$v.json 2> gpurun_out/c5_bench_$v.err
  timeout 200 python bench.py --no-cpu-baseline --no-extras --no-roofline > gpurun_out/c5_bench2_$v.json 2>> gpurun_out/c5_bench_$v.err
done
paste <(grep "^fwd" gpurun_out/c5_kbench_base.txt | cut -c1-100) <(grep "^fwd" gpurun_out/c5_kbench_pin1.txt | cut -c30-100)
for v in base pin1; do python -c "
import json
for f in ('gpurun_out/c5_bench_$v.json','gpurun_out/c5_bench2_$v.json'):
    d=json.load(open(f)); print('$v', d['value'], d.get('roofline',{}).get('kernels'))"; done
