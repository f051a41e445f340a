#!/bin/bash
# round 5, session f: kernel-time breakdown of the 64- and 128-pair steps (rocprofv3 kernel trace), GEMM tests after the choose() change.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -x -k "gemm" > gpurun_out/r05f_pytest.log 2>&1
echo "pytest exit $?"; tail -n 8 gpurun_out/r05f_pytest.log
for b in 64 128; do
  UNIVL_AB= timeout 300 python bench.py --child --batch $b --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('batch $b ms/step', j['ms_per_step'], 'pairs/s', j['value'])"
  (cd /tmp && rm -rf /tmp/prof$b && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$b -o r --output-format csv -- python "$OLDPWD/bench.py" --child --batch $b --steps 10 --warmup 3 --no-graph > /tmp/prof$b.log 2>&1; echo "rocprof exit $?")
  f=$(find /tmp/prof$b -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/r05f_bench_b${b}_kernel_stats.csv
  head -16 "$f" | cut -c1-200
done
