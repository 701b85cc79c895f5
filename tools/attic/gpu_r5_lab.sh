#!/bin/bash
# round 5: standalone kernel labs (no torch): ~15 s of box time per binary
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5lab; mkdir -p $O
for b in "$@"; do
  echo "== $b"; timeout 120 tools/_lab/$b 2>&1 | tee $O/$b.txt
done
