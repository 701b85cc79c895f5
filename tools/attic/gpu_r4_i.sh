#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4i; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bc_one_launch.py -q -x -k "dw or mlp_fwd_bwd or one_launch" > $O/t0.log 2>&1; tail -3 $O/t0.log
echo "== ring (shipped)"; timeout 300 python tools/dw_bench4.py 2>>$O/err.log | tee $O/dw_ring.txt
echo "== no ring"; OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_noring.so timeout 300 python tools/dw_bench4.py 2>>$O/err.log | tee $O/dw_noring.txt
B="timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline --no-cold"
for rep in 1 2 3; do
echo -n "ring   "; $B 2>>$O/err.log | cut -c1-60
echo -n "noring "; OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_noring.so $B 2>>$O/err.log | cut -c1-60
done
echo -n "c4 ring   "; $B --config c4 2>>$O/err.log | cut -c1-60
echo -n "c4 noring "; OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_noring.so $B --config c4 2>>$O/err.log | cut -c1-60
echo -n "c3 ring   "; $B --config c3 2>>$O/err.log | cut -c1-60
echo -n "c3 noring "; OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_noring.so $B --config c3 2>>$O/err.log | cut -c1-60
