cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c39; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "mlp" > $O/t.log 2>&1; tail -5 $O/t.log
timeout 300 python tools/kbench.py 2>&1 | grep "rows=  2048 tile=16" | sed 's/ | /\n    /g'
