#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "big_rows or mlp_fwd" 2>&1 | tail -8
timeout 300 python tools/kbench.py --big 2>&1 | grep "^fwd" | cut -c1-90
for cfg in "20480 80 1 400 8" "20480 80 2 256 1"; do echo "# $cfg"; timeout 60 tools/mlp_phase.bin $cfg 2>&1 | grep -v "^wave" | cut -c1-400; done
for t in 0 80; do OSRL_OOD_TILE=$t timeout 120 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('tile $t steps/s', d['value'], d['roofline']['kernels'], d['last_stats'])"; done
OSRL_ENC_TILE=80 timeout 120 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('enc80 only steps/s', d['value'])"
