#!/bin/bash
# the full GPU suite as the driver runs it at round end (+ durations of the slowest tests)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6suite; rm -rf $O; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
cp gpurun_out/parity_margins.txt $O/ 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
