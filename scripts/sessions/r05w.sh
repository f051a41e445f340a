#!/bin/bash
# Round 5, session w: chunks of cross layer 0 / decoder layer 0 riding in the last text / video layer (tail_ride) -- identity tests, A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05w
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_model_gpu.py -q -x -k "riding or unchanged or pipelined" -p no:cacheprovider 2>&1 | grep -v "Extension modules" | tail -8 | cut -c1-250 | tee $OUT/pytest_riding.txt
b() { local tag=$1; shift; local ab=$1; shift
  UNIVL_AB="$ab" timeout 150 python3 bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-others --no-extras --no-preheat "$@" 2>$OUT/err_$tag.txt | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/ab_tail_ride.txt; grep -v "Extension modules\|amdgpu.ids" $OUT/err_$tag.txt | tail -3 | cut -c1-200; }
for rep in 1 2; do
  b "caption_front_$rep" "tail_ride=0" --kind caption
  b "caption_tail_$rep" "" --kind caption
  b "pretrain_front_$rep" "tail_ride=0" --kind pretrain --batch 6
  b "pretrain_tail_$rep" "" --kind pretrain --batch 6
  b "align_front_$rep" "tail_ride=0" --kind align
  b "align_tail_$rep" "" --kind align
done
