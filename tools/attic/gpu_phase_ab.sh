#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/phase_ab; rm -rf $O; mkdir -p $O
for shape in "256 0 1 256 2" "2048 0 4 256 1" "2048 0 1 400 12"; do
  for cold in 1 0; do
    for tag in r3w0 r4w0 r6w0 r3w1 r4w1; do
      echo "== $tag cold=$cold shape=$shape"
      COLD=$cold timeout 60 ./tools/mlp_phase_$tag.bin $shape 2>&1 | grep -E "mean cycles|kernel span" | cut -c1-400
    done
  done
done > $O/phase_ab.txt 2>&1
cat $O/phase_ab.txt
