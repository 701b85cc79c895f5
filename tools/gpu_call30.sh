cd "$GRAFT_REPO_ROOT"; O=gpurun_out/c30; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "quantile or ood_stat" > $O/t.log 2>&1; tail -5 $O/t.log
timeout 200 python tools/api_probe.py > $O/api.txt 2>&1; cat $O/api.txt | tail -9
timeout 200 python tools/kbench.py 2>&1 | grep -i quantile
timeout 300 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2>$O/bench.err; cut -c1-400 $O/bench.json
timeout 300 python -m pytest tests/test_gpu_train_step.py -q -k "cpq or lazy" > $O/t2.log 2>&1; tail -3 $O/t2.log
