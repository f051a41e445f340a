#!/bin/bash
# Round 4, session c: (1) 768-row products: forced split-K and half-width tiles in the phase trace; (2) data-parallel step with bucket
# norms on the communication stream + captured bf16 exchange (dry-run schedule and world-size-1 RCCL); (3) the new DP test + pruned GEMM tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04c
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 300 python3 scripts/mb_trace_gemm.py --rows 768 > $OUT/trace_gemm_768.txt 2>&1; stamp "trace rc=$?"
grep -E "FORCED|tile|^fwd|^dgrad|pair .*\[all\]" $OUT/trace_gemm_768.txt | cut -c1-200
for b in 4 16; do
  for norms in 1 0; do
    env UNIVL_DP_DRYRUN=1 UNIVL_DP_BUCKET_NORMS=$norms timeout 120 python3 bench.py --force-dp --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-others --no-extras > $OUT/dp_dry_b${b}_n$norms.json 2> $OUT/dp_dry_b${b}_n$norms.err
    echo "dp dry-run b$b bucket-norms=$norms rc=$?: $(grep -o '"ms_per_step": [0-9.]*' $OUT/dp_dry_b${b}_n$norms.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/dp_dry_b${b}_n$norms.json)" | tee -a $OUT/summary.txt
  done
  timeout 120 python3 bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-others --no-extras > $OUT/single_b$b.json 2>/dev/null
  echo "single b$b: $(grep -o '"ms_per_step": [0-9.]*' $OUT/single_b$b.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/single_b$b.json)" | tee -a $OUT/summary.txt
done
for ex in fp32 bf16; do
  env UNIVL_GRAD_EXCHANGE=$ex timeout 120 python3 bench.py --force-dp --steps 50 --warmup 10 --no-cpu-baseline --no-others > $OUT/dp_rccl_b4_$ex.json 2> $OUT/dp_rccl_b4_$ex.err
  echo "dp rccl-world-1 b4 exchange=$ex rc=$?: $(grep -o '"ms_per_step": [0-9.]*' $OUT/dp_rccl_b4_$ex.json | head -1) $(grep -o '"exposed_ms": [0-9.]*' $OUT/dp_rccl_b4_$ex.json) $(grep -o '"collective_ms": [0-9.]*' $OUT/dp_rccl_b4_$ex.json)" | tee -a $OUT/summary.txt
done
stamp "dp done"
timeout 600 python3 -m pytest tests/test_ddp_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest_ddp.log 2>&1; tail -4 $OUT/pytest_ddp.log
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm or misc or colsum or head" > $OUT/pytest_gemm.log 2>&1; tail -4 $OUT/pytest_gemm.log
stamp "done"
