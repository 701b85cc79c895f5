#!/bin/bash
# round 6, first lease: (1) compute-partition probe, READ-ONLY (can one MI355X present its XCDs as RCCL
# ranks?  switching the partition mode is refused by the lease itself -- the refusal text is in DESIGN_LOG.md round 6),
# (2) the new parity tests (bench-path c1 / c5, kink check, counted slab sum, vae_ns widths, random-tuple gradient gate),
# (3) the driver's command on this box as the round's baseline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6a; rm -rf $O; mkdir -p $O
{
  echo "== id"; id
  echo "== rocm-smi --showcomputepartition --showmemorypartition"; rocm-smi --showcomputepartition --showmemorypartition 2>&1
  echo "== amd-smi partition"; (amd-smi partition 2>&1 || true) | head -40
  echo "== after"; rocm-smi --showcomputepartition 2>&1
  echo "== devices"; python -c "import torch; print('torch device_count', torch.cuda.device_count())"; rocminfo 2>/dev/null | grep -c "gfx950" 
  echo "== sysfs"; ls /sys/class/drm/ 2>&1 | head; for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition; do echo $f; cat $f 2>&1; ls -la $f 2>&1; done
} > $O/cpx_probe.txt 2>&1
tail -40 $O/cpx_probe.txt
timeout 1500 python -m pytest tests/test_gpu_bench_path.py "tests/test_gpu_cdt.py::test_counted_slab_sum_equals_full_slab_sum_on_a_real_step" "tests/test_gpu_kernels.py::test_vae_ns_launches_equal_the_fused_launches" "tests/test_gpu_train_step.py::test_random_shape_tuples_match_the_oracle" "tests/test_gpu_cdt.py::test_cdt_c5_full_batch_forward_stats_and_graph" -x -q --durations=8 > $O/pytest_new.log 2>&1; tail -30 $O/pytest_new.log
cp gpurun_out/parity_margins.txt $O/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>$O/bench.err; python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd', d['value'], d['no_preroll'], d['roofline']['frac'], {k:v.get('steps_per_s') for k,v in d['other_configs'].items()})" $O/bench_driver_cmd.json
