#!/bin/bash
# A/B of two builds of the library on one box: osrl_amd/lib/libosrl_amd.so (the tree) vs osrl_amd/lib/libosrl_base.so
# (built from HEAD's sources), via OSRL_LIB (osrl_amd/_lib.py).  Kernel parity tests first, then the per-phase
# cycles of the N*B-row forward kernel, the isolated kernel timings, the headline bench alternating A/B, a timeline.
#   gpurun -- 'bash tools/gpu_lib_ab.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ab; rm -rf $O; mkdir -p $O
BASE=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_base.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py tests/test_gpu_dp_sim.py -q -x -k "not cdt" > $O/t.log 2>&1
grep -E "passed|failed|Error|assert" $O/t.log | tail -8
for b in mlp_phase_base mlp_phase; do
  echo "== $b enc"; timeout 60 ./tools/$b.bin 20480 80 1 400 8 2>&1 | grep -E "mean cycles|kernel span"
  echo "== $b qc"; timeout 60 ./tools/$b.bin 20480 80 2 256 1 2>&1 | grep -E "mean cycles|kernel span"
done
echo "== kbench base"; OSRL_LIB=$BASE timeout 200 python tools/kbench.py --glue 2>&1 | tail -25
echo "== kbench new"; timeout 200 python tools/kbench.py --glue 2>&1 | tail -25
B="timeout 300 python bench.py --no-extras --no-cpu-baseline"
for rep in 1 2; do
  echo "base"; OSRL_LIB=$BASE $B 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('in_step_us'))"
  echo "new"; $B 2>>$O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('in_step_us'))"
done
cd /tmp && rocprofv3 --kernel-trace -f csv -d $O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --no-roofline --steps 200 > $O/bench_profiled.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
T=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
rm -rf $O/prof
cat $O/timeline.txt
