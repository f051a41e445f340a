#!/bin/bash
# Round 3, last session at HEAD: the GPU suite exactly as the driver runs it (one process, -x), the three PMC passes of the step at
# 4 pairs (-> profiles/r03_gemm_pmc.json, stamped with the kernel sources), then the bench line that reads roofline.traffic from it.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-1000}
OUT=gpurun_out/r03y
mkdir -p $OUT
P=$PWD
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
t=$(lim 800)
timeout $t python -m pytest tests/ -x -q -m gpu --durations=15 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_gpu.log | tail -8
stamp "pytest done"
t=$(lim 70); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/$OUT/pmc_fetch --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_fetch.log 2>&1)
t=$(lim 70); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/$OUT/pmc_write --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_write.log 2>&1)
t=$(lim 70); [ $t -gt 25 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/$OUT/pmc_mfma --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_mfma.log 2>&1)
EL=$(grep -o "[0-9]* flat elements" $OUT/pmc_fetch.log | grep -o "^[0-9]*")
python scripts/pmc_step_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma ${EL:-153784064} 4 $OUT/gemm_pmc.json > $OUT/pmc_parse.log 2>&1
cp $OUT/gemm_pmc.json profiles/r03_gemm_pmc.json
for k in fetch write mfma; do find $OUT/pmc_$k -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_$k.csv.gz; rm -rf $OUT/pmc_$k; done
stamp "pmc done"
t=$(lim 150); [ $t -gt 0 ] && { timeout $t python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json; grep -o '"traffic": [0-9.a-z]*' $OUT/bench.json; }
t=$(lim 60); [ $t -gt 20 ] && { timeout $t python bench.py --batch 128 --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $OUT/bench_b128.json 2> $OUT/bench_b128.err; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128.json; }
t=$(lim 60); [ $t -gt 20 ] && { timeout $t python bench.py --batch 16 --steps 150 --warmup 15 --no-cpu-baseline --no-extras > $OUT/bench_b16.json 2> $OUT/bench_b16.err; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b16.json; }
stamp "end"
