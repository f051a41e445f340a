#!/bin/bash
# Round 4, session j: measurements that bound the two fusions VERDICT r3 asks about -- (1) the step with the encoder LayerNorm launches
# left out (UNIVL_PROBE_NO_LN: upper bound of any LayerNorm fusion), (2) non-temporal LDS-DMA of the weight operand in the phase trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04j
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 200 python3 scripts/mb_trace_gemm.py --rows 192 > $OUT/trace_gemm.txt 2>&1
grep -E "^fwd" $OUT/trace_gemm.txt | cut -c1-200
stamp "trace"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 120 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.a-z]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; }
for r in 1 2; do
  line b4_default_$r "X=0" --steps 150 --warmup 10
  line b4_no_ln_fwd_$r "UNIVL_PROBE_NO_LN=fwd" --steps 150 --warmup 10
  line b4_no_ln_bwd_$r "UNIVL_PROBE_NO_LN=bwd" --steps 150 --warmup 10
  line b4_no_ln_both_$r "UNIVL_PROBE_NO_LN=both" --steps 150 --warmup 10
  line b4_nopipe_$r "X=0" --steps 150 --warmup 10 --no-pipeline
  line b16_default_$r "X=0" --batch 16 --steps 100 --warmup 10
  line b16_no_ln_both_$r "UNIVL_PROBE_NO_LN=both" --batch 16 --steps 100 --warmup 10
done
stamp "done"
