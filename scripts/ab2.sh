#!/bin/bash
# within-session A/B of UNIVL_AB variants through `bench.py --child` (own pre-heat, no side measurements): scripts/ab2.sh OUT "ab1" "ab2" ...
# ("" = defaults); 3 interleaved rounds; BENCH_ARGS adds arguments (e.g. "--batch 16").
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
OUT=$1; shift
ARGS="--child --steps 100 --warmup 10 ${BENCH_ARGS:-}"
for r in 1 2 3; do
  for v in "$@"; do
    ms=$(UNIVL_AB="$v" timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "round $r [${v:-default}] ${BENCH_ARGS:-} ms/step: $ms" | tee -a $OUT
  done
done
