#!/bin/bash
# Round 3, session k: bench lines of the other forward() branches (FT-Align, cfg4 caption, cfg5 pretrain) with their own rooflines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${OUTDIR:-r03k}
mkdir -p $OUT
run() { local name=$1; shift
  timeout 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"family_ms_per_step": [0-9.]*' $OUT/bench_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; tail -2 $OUT/bench_$name.err | grep -v amdgpu.ids; }
run joint --kind joint
run align --kind align
run caption --kind caption
run pretrain --kind pretrain --batch 6
run caption_nopipe --kind caption --no-pipeline --no-extras
run pretrain_nopipe --kind pretrain --batch 6 --no-pipeline --no-extras
