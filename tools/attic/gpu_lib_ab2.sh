#!/bin/bash
# A/B of the tree's library against osrl_amd/lib/$1 (OSRL_LIB) on the c2 / c3 / c4 benches, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
ALT=$GRAFT_REPO_ROOT/osrl_amd/lib/$1
O=$GRAFT_REPO_ROOT/gpurun_out/ab2; rm -rf $O; mkdir -p $O
timeout 300 env OSRL_LIB=$ALT python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp or nb" 2>&1 | tail -1
for cfg in ${CFGS:-c2 c3 c4}; do
B="timeout 300 python bench.py --config $cfg --no-extras --no-cpu-baseline"
for rep in ${REPS:-1 2}; do
  echo -n "$cfg base "; $B 2>>$O/err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('in_step_us'))"
  echo -n "$cfg alt  "; OSRL_LIB=$ALT $B 2>>$O/err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('in_step_us'))"
done; done
