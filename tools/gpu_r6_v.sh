#!/bin/bash
# round 6: vae_ns.hip compiled with -amdgpu-mfma-vgpr-form (every vae_ns_gen form <= 304 registers: a wave of it fits beside
# a 208-register wave of the N*B-row launch; before: <0,1> 312, <2,2> 308) -- the vae_ns kernel tests, then A/B against the
# previous build at C2 / C4, K = 200, three rounds
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6v; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "vae_ns or ns" 2>&1 | tail -3 | tee $O/pytest.log
run() {  # cfg label env...
  cfg=$1; lab=$2; shift 2
  env "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 20 > $O/b.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], sys.argv[3], d['value'], d['no_preroll']['value'], d['ms_per_step'])" $O/b.json $cfg "$lab"
}
for rep in 1 2 3; do
  for cfg in c2 c4; do
    run $cfg vgpr-form X=0
    run $cfg previous OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_prev.so
  done
done 2>&1 | tee $O/ab.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $O/prof_c4 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config c4 --no-cpu-baseline --no-extras --no-roofline --steps 200 --warmup 3 > $O/bench_profiled_c4.json 2> $O/prof_c4.err)
python tools/trace_summary.py $(find $O/prof_c4 -name "*kernel_trace.csv" | head -1) > $O/trace_summary_c4.txt 2>&1
python tools/timeline_graph.py $(find $O/prof_c4 -name "*kernel_trace.csv" | head -1) 1 > $O/timeline_c4.txt 2>&1
rm -rf $O/prof_c4; head -30 $O/trace_summary_c4.txt | cut -c1-150
