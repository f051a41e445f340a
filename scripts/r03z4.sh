#!/bin/bash
# Round 3: where the bias-gradient column sums go at 128 pairs -- on the chain (element-per-lane kernel, the r03z2 form) or on the
# weight-gradient stream (16-byte kernel); interleaved.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03z4
mkdir -p $OUT
for rep in 1 2; do
for v in "UNIVL_COLSUM_ON_CHAIN=1 UNIVL_COLSUM_VEC=0" "UNIVL_COLSUM_ON_CHAIN=0 UNIVL_COLSUM_VEC=1"; do
  n=$(echo "$v" | tr ' =' '__')
  env $v timeout 12 python bench.py --batch 128 --steps 50 --warmup 8 --no-cpu-baseline --no-extras > $OUT/bench_${n}_$rep.json 2> $OUT/bench_${n}_$rep.err
  echo "$v rep $rep $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_${n}_$rep.json)" | tee -a $OUT/ab_b128.txt
done
done
