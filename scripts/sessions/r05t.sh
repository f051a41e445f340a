#!/bin/bash
# Round 5, session t: from 1536 tokens a layer's optimizer chunks as a launch on a side stream next to the products of the layer in front
# (side_update_min_rows) instead of serially between them -- A/B at 32 / 64 / 128 pairs, the riding-update identity tests with the side
# form forced onto the small test shapes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05t
mkdir -p $OUT
b() { local tag=$1; shift; local ab=$1; shift
  UNIVL_AB="$ab" timeout 150 python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-others --no-extras --no-preheat "$@" 2>$OUT/err_$tag.txt | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/ab_side_update.txt; tail -2 $OUT/err_$tag.txt; }
UNIVL_AB="side_update_min_rows=64" timeout 300 python3 -m pytest tests/test_model_gpu.py -q -x -k "riding or unchanged or pipelined" -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_side_forced.txt
for rep in 1 2; do
  b "b128_serial_$rep" "side_update_min_rows=0" --batch 128
  b "b128_side_$rep" "" --batch 128
  b "b64_serial_$rep" "side_update_min_rows=0" --batch 64
  b "b64_side_$rep" "" --batch 64
  b "b32_serial_$rep" "side_update_min_rows=0" --batch 32
  b "b32_side_$rep" "" --batch 32
done
b "b16_side_forced" "side_update_min_rows=768" --batch 16
b "b16_default" "" --batch 16
