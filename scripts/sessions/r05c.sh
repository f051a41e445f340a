#!/bin/bash
# round 5, session c: the 256 body with the transposed accumulator blocks (16-byte epilogue accesses): parity, microbenchmark at 6144,
# phase trace (trace build), one SQ counter pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -x -k "gemm256" > gpurun_out/r05c_pytest.log 2>&1
echo "pytest exit $?"; tail -n 15 gpurun_out/r05c_pytest.log
timeout 300 python scripts/mb_gemm256.py --rows 6144 --out gpurun_out/r05c_mb_gemm256.json > gpurun_out/r05c_mb_gemm256.txt 2>&1
echo "mb exit $?"; cat gpurun_out/r05c_mb_gemm256.txt
timeout 300 python scripts/mb_trace_gemm256.py --rows 6144 > gpurun_out/r05c_trace_gemm256.txt 2>&1
echo "trace exit $?"; cat gpurun_out/r05c_trace_gemm256.txt
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d /tmp/pmc256 -o r --output-format csv -- python "$OLDPWD/scripts/pmc_gemm256.py" > /tmp/pmc256.log 2>&1; echo "pmc exit $?")
python scripts/pmc_parse_by_kernel.py /tmp/pmc256 --match=gemm256 --match=gemm_kernel > gpurun_out/r05c_pmc_gemm256.txt 2>&1
cat gpurun_out/r05c_pmc_gemm256.txt
