#!/bin/bash
# Round 4, session e: same-box A/B of the pair form (square / rectangular) x mid split-K at 8 / 16 / 32 pairs, interleaved, two rounds;
# kernel stats of the 16-pair step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04e
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 120 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; }
for r in 1 2; do
  for b in 16 32 8; do
    line b${b}_rect_mid_$r "UNIVL_SPLITK_MID_TILES=512" --batch $b --steps 80 --warmup 10
    line b${b}_rect_nomid_$r "UNIVL_SPLITK_MID=0" --batch $b --steps 80 --warmup 10
    line b${b}_square_mid_$r "UNIVL_PAIR_FORM=square UNIVL_SPLITK_MID_TILES=512" --batch $b --steps 80 --warmup 10
    line b${b}_square_nomid_$r "UNIVL_PAIR_FORM=square UNIVL_SPLITK_MID=0" --batch $b --steps 80 --warmup 10
  done
done
stamp "ab done"
P=$PWD
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof16 --output-format csv -- python3 $P/bench.py --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-others --no-extras > $P/$OUT/prof16.log 2>&1)
find $OUT/prof16 -name "*kernel_stats.csv" -exec cp {} $OUT/bench_b16_kernel_stats.csv \; ; rm -rf $OUT/prof16
head -12 $OUT/bench_b16_kernel_stats.csv | cut -c1-160
stamp "done"
