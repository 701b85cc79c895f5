#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/dwT; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --config c2 --no-cpu-baseline --no-extras --no-roofline"
run() { echo -n "$1 : "; env $1 $B 2>>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
run "X=0"
run "OSRL_DW_T_COST=2"
run "OSRL_DW_T_COST=3"
run "OSRL_DW_T_COST=2 OSRL_DW_S_COST=2"
run "OSRL_DW_T_COST=2 OSRL_DW_T_CRITIC=2"
run "OSRL_DW_T_COST=3 OSRL_DW_T_CRITIC=3"
run "OSRL_DW_T_CRITIC=2"
run "X=0"
