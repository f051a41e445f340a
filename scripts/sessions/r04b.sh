#!/bin/bash
# Round 4, session b: (1) in-kernel phase trace of the 192- / 768-row GEMM-family launches (trace build); (2) the data-parallel step with
# the riding update as TWO graphs (forward with riders | backward with collectives): dry-run schedule and real world-size-1 RCCL.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04b
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 300 python3 scripts/mb_trace_gemm.py --rows 192,768 > $OUT/trace_gemm.txt 2>&1; stamp "trace rc=$?"
cat $OUT/trace_gemm.txt | cut -c1-230
for b in 4 16; do
  env UNIVL_DP_DRYRUN=1 timeout 120 python3 bench.py --force-dp --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-others > $OUT/dp_dry_b$b.json 2> $OUT/dp_dry_b$b.err
  echo "dp dry-run b$b rc=$?: $(grep -o '"ms_per_step": [0-9.]*' $OUT/dp_dry_b$b.json | head -1) $(grep -o '"optimizer_riding": [a-z]*' $OUT/dp_dry_b$b.json) $(grep -o '"graph_mode": "[a-z]*"' $OUT/dp_dry_b$b.json)" | tee -a $OUT/summary.txt
  timeout 120 python3 bench.py --force-dp --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-others > $OUT/dp_rccl_b$b.json 2> $OUT/dp_rccl_b$b.err
  echo "dp rccl-world-1 b$b rc=$?: $(grep -o '"ms_per_step": [0-9.]*' $OUT/dp_rccl_b$b.json | head -1) $(grep -o '"optimizer_riding": [a-z]*' $OUT/dp_rccl_b$b.json) $(grep -o '"exposed_ms": [0-9.]*' $OUT/dp_rccl_b$b.json)" | tee -a $OUT/summary.txt
  tail -3 $OUT/dp_rccl_b$b.err
done
env UNIVL_ADAM_RIDE=0 UNIVL_DP_DRYRUN=1 timeout 120 python3 bench.py --force-dp --steps 50 --warmup 10 --no-cpu-baseline --no-others > $OUT/dp_dry_noride_b4.json 2> $OUT/dp_dry_noride_b4.err
echo "dp dry-run b4 NO riders: $(grep -o '"ms_per_step": [0-9.]*' $OUT/dp_dry_noride_b4.json | head -1)" | tee -a $OUT/summary.txt
timeout 120 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-others --no-extras > $OUT/single_b4.json 2>/dev/null
echo "single b4: $(grep -o '"ms_per_step": [0-9.]*' $OUT/single_b4.json | head -1)" | tee -a $OUT/summary.txt
stamp "dp done"
timeout 300 python3 -m pytest tests/test_ddp_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/pytest_ddp.log 2>&1; tail -3 $OUT/pytest_ddp.log
stamp "done"
