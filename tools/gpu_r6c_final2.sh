#!/bin/bash
# round 6, third session, the final tree (C4 at 20 steps per graph): the no-join pipeline tests, two runs of the driver's command (full line), C4 alone,
# un-profiled stamps of C4's 20-step graph
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6cfinal2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ood_rows.py -x -q > $O/pytest.txt 2>&1; tail -n 2 $O/pytest.txt
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_$i.json 2>>$O/bench.err; python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('driver cmd', d['value'], d['no_preroll'], d['config']['steps_per_graph'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['isolated_frac'], {k:(v.get('steps_per_s'), v.get('steps_per_graph')) for k,v in d['other_configs'].items()})" $O/bench_driver_cmd_$i.json; done
timeout 300 python bench.py --config c4 --steps 300 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_c4.json 2>>$O/bench.err; cut -c1-80 $O/bench_c4.json
OSRL_LIB=$GRAFT_REPO_ROOT/osrl_amd/lib/libosrl_trace.so timeout 300 python tools/trace_steps.py c4 0 40 > $O/trace_unprofiled_c4.txt 2>> $O/bench.err
head -2 $O/trace_unprofiled_c4.txt
tail -n 2 $O/bench.err
