#!/bin/bash
# round 6 (second session): the HIP runtime's graph-execution switches nobody had tried (strings of libamdhip64.so:
# DEBUG_CLR_GRAPH_PACKET_CAPTURE, DEBUG_HIP_FORCE_GRAPH_QUEUES, DEBUG_HIP_GRAPH_BATCH_SIZE) against the default, un-profiled
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp OSRL_LAB=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6env; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-extras --no-roofline --steps 300 --warmup 20"
run() {  # $1 = tag, rest = env assignments
  tag=$1; shift
  for cfg in c2 c4 c3; do
    st="--steps 300"; [ $cfg = c3 ] && st="--steps 100"
    env "$@" timeout 300 python bench.py --config $cfg $B $st > $O/b_${cfg}_$tag.json 2> $O/b_${cfg}_$tag.err
    echo "$cfg $tag $(python -c "import json,sys; d=json.loads(open('$O/b_${cfg}_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
  done
}
for r in 1 2; do
  run default_$r X=1
  run nocapture_$r DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run queues2_$r DEBUG_HIP_FORCE_GRAPH_QUEUES=2
  run queues8_$r DEBUG_HIP_FORCE_GRAPH_QUEUES=8
  run batch1_$r DEBUG_HIP_GRAPH_BATCH_SIZE=1
  run batch64_$r DEBUG_HIP_GRAPH_BATCH_SIZE=64
done
# the no-join C2 graph under the same switches (its kernel timeline is shorter under rocprofv3, its clock longer without)
for v in "X=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2"; do
  env $v OSRL_PIPE_DUAL=next OSRL_PIPE_PROLOGUE=early timeout 300 python bench.py --config c2 $B > $O/b_c2_nojoin.json 2> $O/b_c2_nojoin.err
  echo "c2 no-join/early [$v] $(python -c "import json,sys; d=json.loads(open('$O/b_c2_nojoin.json').read().strip().splitlines()[-1]); print(d['value'], d.get('no_preroll'))" 2>&1 | tail -1)" | tee -a $O/ab.txt
done
# what the executor says about its streams (one short run, log level 4 filtered)
AMD_LOG_LEVEL=4 timeout 300 python bench.py --config c2 --no-cpu-baseline --no-extras --no-roofline --steps 20 --warmup 5 2>&1 >/dev/null | grep -i "hipGraph\]\|GraphExec::Run\|max streams\|parallel streams" | sort | uniq -c | sort -rn | head -20 > $O/graph_log.txt
cat $O/graph_log.txt
