#!/bin/bash
# round 4, step C: which cost-critic dW shape co-resides with the 8-wave encoder launch (32 x 32 tiles: 17 KB, 154 registers)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4c; rm -rf $O; mkdir -p $O
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline"
run() { echo "$*"; env "$@" $B 2>>$O/bench.err | cut -c1-60; }
for rep in 1 2; do
run OSRL_NB_WAVES=8
run OSRL_NB_WAVES=8 OSRL_DW_T_COST=2 OSRL_DW_S_COST=2
run OSRL_NB_WAVES=8 OSRL_DW_T_COST=2 OSRL_DW_S_COST=4
run OSRL_NB_WAVES=8 OSRL_DW_T_COST=2 OSRL_DW_S_COST=1
run OSRL_NB_WAVES=8 OSRL_DW_T_COST=2 OSRL_DW_S_COST=2 OSRL_DW_T_CRITIC=2 OSRL_DW_S_CRITIC=2
run OSRL_NB_WAVES=4 OSRL_DW_T_COST=2 OSRL_DW_S_COST=2
done
cd /tmp && OSRL_NB_WAVES=8 OSRL_DW_T_COST=2 OSRL_DW_S_COST=2 rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
rm -rf $O/prof
tail -16 $O/timeline.txt
