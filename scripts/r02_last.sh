#!/bin/bash
# Last GPU session of round 2 (about 9 minutes of box time were left): the full GPU suite as two concurrent pytest
# processes, smoke, the default bench line, the eager kernel trace, then same-session A/B runs of the GEMM switches.
# Every step is bounded by what is left of BUDGET seconds, so the call never runs into gpurun's own limit.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-560}
OUT=gpurun_out/r02p
mkdir -p $OUT
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
import torch
print('preflight ok', torch.cuda.get_device_name(0))" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
nproc >> $OUT/preflight.txt
stamp "preflight done"
# ---- 1. the GPU suite: the model tests (they write gpurun_out/parity_errors.json) beside everything else
t=$(lim 420)
(timeout $t python -m pytest tests/test_model_gpu.py -m gpu -x -q --durations=8 > $OUT/pytest_model.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout $t python -m pytest tests -m gpu -x -q --durations=8 --ignore=tests/test_model_gpu.py > $OUT/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_rest.log) &
P2=$!
wait $P1 $P2
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
tail -3 $OUT/pytest_model.log; tail -3 $OUT/pytest_rest.log
stamp "pytest done"
t=$(lim 120); [ $t -gt 0 ] && { (timeout $t python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log); tail -2 $OUT/smoke.log; }
stamp "smoke done"
# ---- 2. the bench line of this commit
t=$(lim 200); [ $t -gt 0 ] && { timeout $t python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-260 $OUT/bench.json; }
stamp "bench done"
# ---- 3. kernel trace (eager: every kernel its own dispatch), same command as the earlier rounds
P=$PWD
t=$(lim 150); [ $t -gt 0 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/$OUT/prof -o eager --output-format csv -- python $P/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-extras > $P/$OUT/prof_bench.json 2> $P/$OUT/prof_bench.err)
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/eager_kernel_stats.csv \; ; rm -rf $OUT/prof; }
stamp "kernel trace done"
# ---- 4. same-session A/B of the GEMM switches (100 steps each)
ab() {   # name batch env...
  local name=$1 batch=$2 t; shift 2
  t=$(lim 90); [ $t -gt 0 ] || return
  env "$@" timeout $t python bench.py --batch $batch --steps 100 --warmup 15 --no-cpu-baseline --no-extras > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt
}
ab b4_base 4 UNIVL_X=0
ab b4_w4 4 UNIVL_GEMM_WAVES=4
ab b4_base2 4 UNIVL_X=0
ab b16_base 16 UNIVL_X=0
ab b16_gm0 16 UNIVL_GEMM_GM=0
ab b128_base 128 UNIVL_X=0
ab b128_gm0 128 UNIVL_GEMM_GM=0
ab b128_t256 128 UNIVL_GEMM_T256_MIN=1
ab b128_w4 128 UNIVL_GEMM_WAVES=4
t=$(lim 90); [ $t -gt 0 ] && { timeout $t python bench.py --loopback --steps 100 --warmup 15 --no-cpu-baseline --no-extras > $OUT/ab_b4_loopback.json 2> $OUT/ab_b4_loopback.err; echo "b4_loopback: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_b4_loopback.json)" | tee -a $OUT/ab_summary.txt; }
stamp "A/B done"
# ---- 5. whole-step graph under the kernel trace: which nodes one replay really has (copies / fills included)
t=$(lim 120); [ $t -gt 20 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --stats -d $P/$OUT/profg -o graph --output-format csv -- python $P/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $P/$OUT/profg_bench.json 2> $P/$OUT/profg_bench.err)
  find $OUT/profg -name "*kernel_stats.csv" -exec cp {} $OUT/graph_kernel_stats.csv \; ; rm -rf $OUT/profg; }
stamp "graph trace done"
# ---- 6. MFMA counters of the 8-wave kernels (own pass, --kernel-trace only beside --pmc)
t=$(lim 120); [ $t -gt 30 ] && { (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/$OUT/pmc_mfma --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_mfma.log 2>&1)
  find $OUT/pmc_mfma -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_mfma.csv.gz; rm -rf $OUT/pmc_mfma; }
stamp "end"
