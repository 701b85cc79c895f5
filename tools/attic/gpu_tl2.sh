#!/bin/bash
# step timelines (rocprofv3 kernel trace) under several environments:  gpurun -- 'bash tools/gpu_tl2.sh TAG "ENV A" "ENV B" ...'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TAG=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== [$v]"
  (cd /tmp && env $v rocprofv3 --kernel-trace -f csv -d $O/prof$i -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 --config ${CFG:-c2} > $O/bench_profiled$i.json 2> $O/prof$i.err)
  T=$(find $O/prof$i -name "*kernel_trace.csv" | head -1)
  python tools/timeline.py $T > $O/timeline$i.txt 2>&1
  python tools/trace_summary.py $T > $O/summary$i.txt 2>&1
  rm -rf $O/prof$i
  python -c "import json;d=json.loads(open('$O/bench_profiled$i.json').readline());print(d['value'],d['ms_per_step'])"
  cat $O/timeline$i.txt
done
