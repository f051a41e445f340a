#!/bin/bash
# Round 5, session u: optimizer chunks riding in the 64 x 128-tile forward products (gemm_adam_rect_kernel) from 1536 tokens on.
# Identity test, then the same bench command as session t's "serial" lines (4.80 - 4.87 / 7.10 - 7.27 / 11.32 - 11.58 ms at 32 / 64 / 128 pairs).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05u
mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_model_gpu.py -q -x -k "riding or unchanged or pipelined" -p no:cacheprovider 2>&1 | grep -v "Extension modules" | tail -15 | cut -c1-250 | tee $OUT/pytest_riding.txt
b() { local tag=$1; shift; local ab=$1; shift
  UNIVL_AB="$ab" timeout 150 python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-others --no-extras --no-preheat "$@" 2>$OUT/err_$tag.txt | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/ab_rect_rider.txt; grep -v "Extension modules\|amdgpu.ids" $OUT/err_$tag.txt | tail -3 | cut -c1-200; }
for rep in 1 2; do
  b "b128_$rep" "" --batch 128
  b "b64_$rep" "" --batch 64
  b "b32_$rep" "" --batch 32
done
b "b16" "" --batch 16
b "b4" ""
b "caption" "" --kind caption
b "pretrain" "" --kind pretrain --batch 6
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $PWD/prof_u --output-format csv -- python3 $OLDPWD/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-others --no-extras --no-preheat --batch 128 > /dev/null 2>&1; find /tmp/prof_u -name "*kernel_stats.csv" -exec cp {} $OLDPWD/$OUT/bench_b128_kernel_stats.csv \; )
head -12 $OUT/bench_b128_kernel_stats.csv | cut -c1-160
