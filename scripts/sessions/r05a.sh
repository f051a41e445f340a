#!/bin/bash
# round 5, session a: first contact of the 256 x 256 body (csrc/gemm256.h) with the hardware -- parity tests, then the shape-by-shape
# microbenchmark against the older tiles.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -x -k "gemm256 or probe_layouts" > gpurun_out/r05a_pytest.log 2>&1
echo "pytest exit $?"
tail -n 30 gpurun_out/r05a_pytest.log
timeout 400 python scripts/mb_gemm256.py --out gpurun_out/r05a_mb_gemm256.json > gpurun_out/r05a_mb_gemm256.txt 2>&1
echo "mb exit $?"
cat gpurun_out/r05a_mb_gemm256.txt
