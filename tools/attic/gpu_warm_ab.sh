#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/warm; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-extras"
run() { echo -n "$* : "; $B "$@" 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('preroll_ms'))"; }
for r in 1 2 3; do
run --steps 20 --warmup 5
run --steps 20 --warmup 5 --preroll-ms 0
run --steps 20 --warmup 5 --preroll-ms 100
run --steps 200 --warmup 5
done
