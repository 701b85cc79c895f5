#!/bin/bash
# round 6: (1) attention keep hand-off + fused one-output head of the N*B kernel: kernel tests; (2) A/B of both at C2 / C3 / C5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6h; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py::test_mlp_fwd_big_rows tests/test_gpu_cdt.py::test_attention_keep_handoff_is_bit_identical tests/test_gpu_cdt.py::test_attention_wave_splits_padding_and_dropout "tests/test_gpu_cdt.py::test_cdt_dropout_train_step_matches_oracle" tests/test_gpu_pipeline.py::test_steps_replay_follows_the_plan -x -q > $O/pytest.log 2>&1; tail -6 $O/pytest.log
run() {  # cfg steps warm label env...
  cfg=$1; st=$2; wu=$3; lab=$4; shift 4
  env OSRL_LAB=1 "$@" timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-extras --no-roofline --steps $st --warmup $wu > $O/b.json 2>>$O/bench.err
  python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], sys.argv[3], d['value'], d['no_preroll']['value'], d['ms_per_step'])" $O/b.json $cfg "$lab"
}
for rep in 1 2 3; do
  run c3 60 10 head-fused X=0
  run c3 60 10 head-as-layer OSRL_NB_HEAD=0
  run c2 200 20 head-fused X=0
  run c2 200 20 head-as-layer OSRL_NB_HEAD=0
  run c4 200 20 head-fused X=0
  run c4 200 20 head-as-layer OSRL_NB_HEAD=0
  run c5 10 3 keep-handoff X=0
  run c5 10 3 philox-in-backward OSRL_CDT_ATTN_KEEP=0
done 2>&1 | tee $O/ab.txt
tail -3 $O/bench.err
