cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/c33; mkdir -p $O
timeout 200 python tools/begin_bench.py 2>&1 | tail -8
for fb in 0 1; do for cs in 0 1; do echo "FUSED_BEGIN=$fb COST_FWD_SIDE=$cs"; OSRL_FUSED_BEGIN=$fb OSRL_COST_FWD_SIDE=$cs timeout 300 python bench.py --no-extras --no-cpu-baseline --no-roofline 2>>$O/bench.err | cut -c1-60; done; done
