#!/bin/bash
# round 6: the IPC exchange's published buffers as FINE-GRAINED device memory (default now) against the plain allocation
# (OSRL_IPC_COARSE=1): the process tests on the default, then forced data parallelism on one rank and two processes on
# the one GPU, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$GRAFT_REPO_ROOT/gpurun_out/r6u; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_ipc_dp.py -x -q --durations=5 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d.get('dp_exchange'), [ (c['what'], c['us']) for c in (d.get('collectives_in_step') or []) if isinstance(c, dict)])" $1 "$2"; }
for rep in 1 2; do
  for coarse in 0 1; do
    OSRL_IPC_COARSE=$coarse OSRL_FORCE_DP=1 OSRL_DP_EXCHANGE=ipc timeout 300 python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/b.json 2>>$O/bench.err; show $O/b.json "c2 forced DP, IPC, coarse=$coarse"
    OSRL_IPC_COARSE=$coarse OSRL_IPC_ONE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $O/b2.json 2>>$O/bench.err; show $O/b2.json "c2, two processes on one GPU, coarse=$coarse"
  done
done 2>&1 | tee $O/ab.txt
tail -5 $O/bench.err
