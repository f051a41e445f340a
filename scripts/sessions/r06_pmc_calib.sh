#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_pmc_calib
mkdir -p $OUT
P=$PWD
for k in fetch write; do
  c=$( [ $k = fetch ] && echo FETCH_SIZE || echo WRITE_SIZE )
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c -d $P/$OUT/pmc_$k --output-format csv -- python3 $P/scripts/pmc_gemm_calib.py > $P/$OUT/pmc_$k.log 2>&1)
  tail -2 $OUT/pmc_$k.log
done
python3 scripts/pmc_gemm_calib_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/gemm_calib.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write
