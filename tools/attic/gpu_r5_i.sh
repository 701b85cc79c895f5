#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/b.json 2>$O/b.err
python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['no_preroll'], r['kernel'], r['frac'], r['in_step_us'], r['isolated_us'], r.get('error'))
print({k:(v["in_step_us"], v["in_step_how"], v["stamp_boundary_us"], v["in_step_eager_us"], v["isolated_us"]) for k,v in r['kernels'].items()})"
tail -3 $O/b.err
