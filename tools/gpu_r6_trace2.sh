#!/bin/bash
# round 6 (third session): un-profiled kernel start stamps of the C2 / C4 graphs with plan.ood_rows (select / sum / polyak stamped)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6trace2; rm -rf $O; mkdir -p $O
for cfg in c2 c4; do
  OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_trace.so timeout 300 python tools/trace_steps.py $cfg 0 40 > $O/trace_unprofiled_$cfg.txt 2>> $O/err.txt
  head -2 $O/trace_unprofiled_$cfg.txt
done
tail -n 3 $O/err.txt
