#!/bin/bash
# Round 3, last session: the 128-pair plan with the big-tile grouped weight gradients + column-sum bias gradients
# (engine.EncoderStack, UNIVL_WGRAD_BIG_MIN: 2048 at the time of this session, tied to the no-pair regime since) -- the 128-pair parity tests, an interleaved A/B against the former plan, the
# data-parallel form, and a kernel trace of the new plan.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-95}
OUT=gpurun_out/r03z2
mkdir -p $OUT
P=$PWD
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 40); timeout $t python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "b128" -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
stamp "tests done"
for rep in 1 2; do
for v in "UNIVL_WGRAD_BIG_MIN=100000000" "UNIVL_WGRAD_BIG_MIN=2048"; do
  t=$(lim 20); [ $t -gt 9 ] || break
  env $v timeout $t python bench.py --batch 128 --steps 50 --warmup 8 --no-cpu-baseline --no-extras > $OUT/bench_b128_${v}_$rep.json 2> $OUT/bench_b128_${v}_$rep.err
  echo "$v rep $rep $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128_${v}_$rep.json) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_b128_${v}_$rep.json)" | tee -a $OUT/ab_b128.txt
done
done
stamp "ab done"
t=$(lim 20); [ $t -gt 9 ] && { timeout $t python bench.py --batch 128 --steps 40 --warmup 8 --force-dp --no-cpu-baseline --no-extras > $OUT/bench_b128_dp.json 2> $OUT/bench_b128_dp.err; echo "dp b128 $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128_dp.json)" | tee -a $OUT/ab_b128.txt; }
t=$(lim 30); [ $t -gt 12 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/$OUT/prof --output-format csv -- python $P/bench.py --batch 128 --steps 12 --warmup 4 --no-graph --no-cpu-baseline --no-extras > $P/$OUT/prof.log 2>&1)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_b128_kernel_stats.csv \; ; rm -rf $OUT/prof
head -6 $OUT/bench_b128_kernel_stats.csv | cut -c1-200
stamp "end"
