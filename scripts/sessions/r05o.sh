#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r05o; mkdir -p $OUT; P=$PWD
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof_b4 --output-format csv -- python3 $P/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-others --no-extras --no-preheat > $P/$OUT/prof_b4.log 2>&1)
find $OUT/prof_b4 -name "*kernel_stats.csv" -exec cp {} $OUT/bench_b4_kernel_stats.csv \; ; rm -rf $OUT/prof_b4
head -22 $OUT/bench_b4_kernel_stats.csv | cut -c1-210
UNIVL_AB=stamps=1 timeout 120 python scripts/probe_branches.py --batch 4 --steps 60 2>&1 | tail -25
