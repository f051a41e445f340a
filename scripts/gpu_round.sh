#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${1:-all}"
echo "== $(date) host: $(nproc) cpus; $(rocminfo 2>/dev/null | grep -m1 'Marketing Name.*MI' || true)" | tee gpurun_out/session.log
if [[ "$WHAT" == all || "$WHAT" == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --no-header -rfE -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a gpurun_out/session.log
  tail -n 60 gpurun_out/pytest_gpu.log
fi
if [[ "$WHAT" == all || "$WHAT" == smoke ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/session.log
  tail -n 5 gpurun_out/smoke.log
fi
if [[ "$WHAT" == all || "$WHAT" == bench ]]; then
  timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" | tee -a gpurun_out/session.log
  cat gpurun_out/bench.json; tail -n 5 gpurun_out/bench.err
  timeout 300 python bench.py --steps 30 --warmup 5 --no-graph --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err; cat gpurun_out/bench_eager.json
  timeout 300 python bench.py --steps 20 --warmup 5 --batch 16 --no-cpu-baseline > gpurun_out/bench_b16.json 2> gpurun_out/bench_b16.err; cat gpurun_out/bench_b16.json
  timeout 300 python bench.py --steps 10 --warmup 3 --batch 128 --no-cpu-baseline > gpurun_out/bench_b128.json 2> gpurun_out/bench_b128.err; cat gpurun_out/bench_b128.json
fi
if [[ "$WHAT" == all || "$WHAT" == prof ]]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r --output-format csv -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-graph --no-cpu-baseline > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err"); echo "rocprof exit $?" | tee -a gpurun_out/session.log
  find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -n 25
fi
ls -la gpurun_out | tail -n 20
