#!/bin/bash
# round 6: IPC exchange -- world-size sweep of the process test, forced data parallelism on one rank (RCCL vs IPC), two
# PROCESSES on the one GPU (collectives' in-step cost with a real peer)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6l; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ipc_dp.py -x -q --durations=8 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d.get('dp_exchange'), [ (c['what'], c['us']) for c in (d.get('collectives_in_step') or []) if isinstance(c, dict)], (d.get('other_configs') or {}).get('c4'))" $1 "$2"; }
for rep in 1 2; do
  timeout 300 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/c2_single.json 2>>$O/bench.err; show $O/c2_single.json "c2 single plan"
  OSRL_FORCE_DP=1 timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/c2_forced_rccl.json 2>>$O/bench.err; show $O/c2_forced_rccl.json "c2 forced DP, RCCL"
  OSRL_FORCE_DP=1 OSRL_DP_EXCHANGE=ipc timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $O/c2_forced_ipc.json 2>>$O/bench.err; show $O/c2_forced_ipc.json "c2 forced DP, IPC"
  timeout 300 python bench.py --config c4 --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/c4_single.json 2>>$O/bench.err; show $O/c4_single.json "c4 single plan"
done 2>&1 | tee $O/forced.txt
OSRL_IPC_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline > $O/c2_two_procs_one_gpu.json 2>>$O/bench.err; show $O/c2_two_procs_one_gpu.json "c2, two processes on one GPU (IPC)" | tee -a $O/forced.txt
tail -5 $O/bench.err
