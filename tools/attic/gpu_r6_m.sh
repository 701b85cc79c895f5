#!/bin/bash
# round 6: IPC exchange (write-through stores / system-scope loads, the rank's slab sum fused into the publish): process
# tests, then forced data parallelism on one rank against RCCL's no-op collectives and the single plan
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6m; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ipc_dp.py -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], [ (c['what'], c['us']) for c in (d.get('collectives_in_step') or []) if isinstance(c, dict)])" $1 "$2"; }
for rep in 1 2; do
  for w in 64; do
    OSRL_IPC_WGS=$w OSRL_FORCE_DP=1 OSRL_DP_EXCHANGE=ipc timeout 300 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/b.json 2>>$O/bench.err; show $O/b.json "c2 forced DP, IPC, $w workgroups"
  done
  OSRL_FORCE_DP=1 timeout 300 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/b.json 2>>$O/bench.err; show $O/b.json "c2 forced DP, RCCL"
  timeout 300 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline --steps-per-graph 1 > $O/b.json 2>>$O/bench.err; show $O/b.json "c2 single plan, one step per graph"
done 2>&1 | tee $O/sweep.txt
