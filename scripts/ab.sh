#!/bin/bash
# within-session A/B of env-selected variants: scripts/ab.sh "VAR=a VAR2=b" "VAR=c" ...  (3 interleaved rounds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
ARGS="${BENCH_ARGS:---steps 40 --warmup 10 --no-cpu-baseline}"
for r in 1 2 3; do
  for v in "$@"; do
    out=$(env $v timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'])")
    echo "round $r [$v] ms/step pairs/s adam_ms: $out" | tee -a gpurun_out/ab.log
  done
done
