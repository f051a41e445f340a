#!/bin/bash
# Round 6, final HEAD: what r06_final5.sh left out -- rocprofv3 --kernel-trace --stats at 16 / 128 pairs, caption, pretrain; device stamps of the
# 4-pair step; the N > 1 code path at world size 1 with the cfg3 row (bench.py --force-dp --cfg3-row).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_final5
mkdir -p $OUT
P=$PWD
prof() { local name=$1; shift
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof_$name --output-format csv -- python3 $P/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-others --no-extras --no-preheat "$@" > $P/$OUT/prof_$name.log 2>&1)
  find $OUT/prof_$name -name "*kernel_stats.csv" -exec cp {} $OUT/bench_${name}_kernel_stats.csv \; ; rm -rf $OUT/prof_$name
  head -3 $OUT/bench_${name}_kernel_stats.csv | cut -c1-150; }
prof b16 --batch 16
prof b128 --batch 128 --steps 6
prof caption --kind caption
prof pretrain --kind pretrain --batch 6
UNIVL_AB=stamps=1 timeout 120 python3 scripts/probe_branches.py > $OUT/probe_stamps_b4.txt 2>&1; tail -14 $OUT/probe_stamps_b4.txt
timeout 300 python3 bench.py --force-dp --cfg3-row --no-others --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_force_dp_cfg3_row.json 2> $OUT/bench_force_dp.err; echo "force-dp rc=$?"; cut -c1-300 $OUT/bench_force_dp_cfg3_row.json
