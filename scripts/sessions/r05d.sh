#!/bin/bash
# round 5, session d: K rotation A/B (libunivl_hip.so vs the -DG256_KROT=0 build), L2 counters of the 256 body.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -rfE -p no:cacheprovider -x -k "gemm256" > gpurun_out/r05d_pytest.log 2>&1
echo "pytest exit $?"; tail -n 5 gpurun_out/r05d_pytest.log
for lib in "" norot; do
  echo "== lib: ${lib:-default (K rotation)}"
  UNIVL_LIB=${lib:+$PWD/univl_amd/lib/libunivl_hip_$lib.so} timeout 300 python scripts/mb_gemm256.py --rows 6144 --kinds fwd,dgrad,group --out gpurun_out/r05d_mb_${lib:-rot}.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05d_mb_${lib:-rot}.txt
done
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|TCP_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/r05d_counter_names.txt
for lib in "" norot; do
  (cd /tmp && rm -rf /tmp/pmcT && UNIVL_LIB=${lib:+$OLDPWD/univl_amd/lib/libunivl_hip_$lib.so} timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d /tmp/pmcT -o r --output-format csv -- python "$OLDPWD/scripts/pmc_gemm256.py" > /tmp/pmcT.log 2>&1; echo "pmc exit $?"; tail -3 /tmp/pmcT.log)
  echo "== TCC counters, lib: ${lib:-default (K rotation)}" | tee -a gpurun_out/r05d_pmc_tcc.txt
  python scripts/pmc_parse_by_kernel.py /tmp/pmcT --match=gemm256 --match=gemm_kernel | tee -a gpurun_out/r05d_pmc_tcc.txt
done
