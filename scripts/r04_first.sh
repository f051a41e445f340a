#!/bin/bash
# PREPARED for round 4 (not run yet): first session at the round-3 HEAD.  A session of the last kind costs 20-90 s of box time
# (box acquisition + push ~10-35 s), so this one is cut into three independent parts -- run them as separate gpurun calls:
#   bash scripts/r04_first.sh suite     driver-form GPU suite (~375 s), smoke
#   bash scripts/r04_first.sh lines     bench lines at 4 / 16 / 64 / 128 pairs + the other forward() branches, kernel traces at 4 and
#                                       128 pairs, the three PMC passes (stamp), the stall pass at 128 pairs on the new plan (~200 s)
#   bash scripts/r04_first.sh mid       what round 3 left unmeasured: the 2048 .. 5461-token regime (32 / 64 / 96 pairs) with the
#                                       big-tile weight-gradient plan forced on / off, half-width tiles on / off there, and the
#                                       per-shape variant tables at 1536 / 3072 rows (~120 s)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PART=${1:-lines}
T0=$(date +%s)
BUDGET=${BUDGET:-420}
OUT=gpurun_out/r04a_$PART
mkdir -p $OUT
P=$PWD
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
line() {   # name "ENV=.. ENV=.." bench args...
  local name=$1 envs=$2 t; shift 2
  t=$(lim 40); [ $t -gt 10 ] || return
  env $envs timeout $t python bench.py --no-cpu-baseline --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt
}
if [ "$PART" = suite ]; then
  t=$(lim 400)
  timeout $t python -m pytest tests/ -x -q -m gpu --durations=15 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
  grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_gpu.log | tail -8
  t=$(lim 40); [ $t -gt 10 ] && { timeout $t python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log; }
  stamp "suite done"
elif [ "$PART" = lines ]; then
  for k in fetch write; do
    c=$( [ $k = fetch ] && echo FETCH_SIZE || echo WRITE_SIZE )
    t=$(lim 40); [ $t -gt 15 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc $c -d $P/$OUT/pmc_$k --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_$k.log 2>&1)
  done
  t=$(lim 40); [ $t -gt 15 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/$OUT/pmc_mfma --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_mfma.log 2>&1)
  EL=$(grep -o "[0-9]* flat elements" $OUT/pmc_fetch.log | grep -o "^[0-9]*")
  python scripts/pmc_step_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma ${EL:-153784064} 4 $OUT/gemm_pmc.json > $OUT/pmc_parse.log 2>&1   # copy to profiles/r0N_gemm_pmc.json by hand (gpurun merges gpurun_out/ only)
  for k in fetch write mfma; do find $OUT/pmc_$k -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_$k.csv.gz; rm -rf $OUT/pmc_$k; done
  stamp "pmc done"
  t=$(lim 90); [ $t -gt 30 ] && { timeout $t python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-220 $OUT/bench.json; }
  line b4 "UNIVL_X=0" --steps 150 --warmup 15
  line b16 "UNIVL_X=0" --batch 16 --steps 150 --warmup 15
  line b64 "UNIVL_X=0" --batch 64 --steps 60 --warmup 8
  line b128 "UNIVL_X=0" --batch 128 --steps 60 --warmup 8
  line b128_dp "UNIVL_X=0" --batch 128 --steps 40 --warmup 8 --force-dp
  for k in align caption pretrain; do line kind_$k "UNIVL_X=0" --kind $k --steps 60 --warmup 10; done
  stamp "bench lines done"
  for b in 4 128; do
    t=$(lim 40); [ $t -gt 15 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/$OUT/prof$b --output-format csv -- python $P/bench.py --batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $P/$OUT/prof$b.log 2>&1)
      find $OUT/prof$b -name "*kernel_stats.csv" -exec cp {} $OUT/bench_b${b}_graph_kernel_stats.csv \; ; rm -rf $OUT/prof$b; }
  done
  t=$(lim 60); [ $t -gt 25 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $P/$OUT/pmc_stall128 --output-format csv -- python $P/scripts/pmc_step.py 128 > $P/$OUT/pmc_stall128.log 2>&1)
    python scripts/pmc_stall_parse.py $OUT/pmc_stall128 > $OUT/pmc_stall_b128.txt 2>&1; find $OUT/pmc_stall128 -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_stall128.csv.gz; rm -rf $OUT/pmc_stall128; head -16 $OUT/pmc_stall_b128.txt; }
  stamp "traces done"
else
  t=$(lim 30); timeout $t python scripts/mb_gemm_variants.py --rows 1536,3072 --kinds fwd,dgrad --out $OUT/mb_mid.json > $OUT/mb_mid.txt 2>&1
  for T in 1536 3072; do
    t=$(lim 20); timeout $t python scripts/mb_gemm_variants.py --rows "" --group-rows $T --group-dbias 0 --out $OUT/mb_group_${T}_nodbias.json > $OUT/mb_group_${T}_nodbias.txt 2>&1
    t=$(lim 20); timeout $t python scripts/mb_gemm_variants.py --rows "" --group-rows $T --group-dbias 1 --out $OUT/mb_group_${T}.json > $OUT/mb_group_${T}.txt 2>&1
  done
  stamp "tables done"
  for b in 32 64 96; do
    for rep in 1 2; do
      line b${b}_default_$rep "UNIVL_X=0" --batch $b --steps 60 --warmup 8
      line b${b}_bigwgrad_$rep "UNIVL_WGRAD_BIG_MIN=1024" --batch $b --steps 60 --warmup 8
      line b${b}_norect_$rep "UNIVL_GEMM_RECT=0" --batch $b --steps 60 --warmup 8
    done
  done
  stamp "mid regime done"
fi
stamp "end"
