#!/bin/bash
# round 3, call A: does the order "eager probe -> capture" cost the two-branch graph its overlap?  + lease facts
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3a; rm -rf $O; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.readline()); r=d.get("roofline") or {}; print(d["value"], d["ms_per_step"], r.get("isolated_us"), r.get("in_step_us"))'
{
echo "== lease facts"
rocm-smi --showclocks --showpower --showcomputepartition --showmemorypartition --showdriverversion --showperflevel 2>&1 | head -60
cat /sys/module/amdgpu/version 2>&1
uname -r
env | grep -E '^(HSA_|GPU_|HIP_|ROC|AMD_|NCCL|RCCL)' | sort
ls /sys/class/drm/ 2>&1 | head
cat /sys/class/drm/card*/device/current_compute_partition 2>&1
cat /sys/class/drm/card*/device/current_memory_partition 2>&1
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -8
cat /sys/class/drm/card*/device/pp_dpm_mclk 2>&1 | head -8
cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>&1
echo "== driver command (probes before the timed region)"
for rep in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>>$O/bench.err | python -c "$P"; done
echo "== same, --no-roofline (no eager probe before the capture)"
for rep in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline 2>>$O/bench.err | python -c "$P"; done
echo "== long timed region, --no-roofline"
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-roofline 2>>$O/bench.err | python -c "$P"
echo "== re-captures in one process"
timeout 300 python tools/graph_variance.py c2 8 2>>$O/gv.err
echo "== GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/graph_variance.py c2 5 2>>$O/gv.err
echo "== GPU_MAX_HW_QUEUES=2"
GPU_MAX_HW_QUEUES=2 timeout 300 python tools/graph_variance.py c2 4 2>>$O/gv.err
} > $O/out.txt 2>&1
tail -5 $O/bench.err $O/gv.err >> $O/out.txt 2>&1
cat $O/out.txt
