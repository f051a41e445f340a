#!/bin/bash
# Final GPU session of round 2: full GPU suite, smoke, the bench lines, kernel trace, PMC passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02m
mkdir -p $OUT
python -c "
from univl_amd import _lib
L = _lib.lib()
missing = [n for n in _lib.EXPORTED if not hasattr(L, n)]
assert not missing, missing
print('preflight ok')" > $OUT/preflight.txt 2>&1 || { cat $OUT/preflight.txt; exit 7; }
rocminfo | grep -m2 "Marketing Name" > $OUT/box.txt 2>&1; nproc >> $OUT/box.txt
(timeout 900 python -m pytest tests -m gpu -x -q -rs --durations=6 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log)
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
tail -4 $OUT/pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log); tail -2 $OUT/smoke.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
for r in 1 2; do
  UNIVL_GEMM_XCD=0 timeout 300 python bench.py --steps 150 --warmup 20 --no-cpu-baseline > $OUT/bench_xcd0_$r.json 2> $OUT/bench_xcd0_$r.err
  UNIVL_GEMM_XCD=1 timeout 300 python bench.py --steps 150 --warmup 20 --no-cpu-baseline > $OUT/bench_xcd1_$r.json 2> $OUT/bench_xcd1_$r.err
done
UNIVL_GEMM_XCD=0 timeout 300 python bench.py --batch 16 --no-cpu-baseline > $OUT/bench_b16_xcd0.json 2> $OUT/bench_b16_xcd0.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/bench_driver.err
timeout 300 python bench.py --batch 16 --no-cpu-baseline > $OUT/bench_b16.json 2> $OUT/bench_b16.err
timeout 300 python bench.py --batch 128 --no-cpu-baseline > $OUT/bench_b128.json 2> $OUT/bench_b128.err
timeout 300 python bench.py --no-graph --no-cpu-baseline --no-extras > $OUT/bench_eager.json 2> $OUT/bench_eager.err
timeout 300 python bench.py --loopback --no-cpu-baseline --no-extras > $OUT/bench_loopback.json 2> $OUT/bench_loopback.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_gpus1.json 2> $OUT/bench_gpus1.err
(timeout 120 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; echo "rc=$?" >> $OUT/bench_gpus2.err)
timeout 300 python bench.py --force-dp --no-cpu-baseline --no-extras > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
timeout 300 python bench.py --force-dp --shard-optimizer --no-cpu-baseline --no-extras > $OUT/bench_rccl_world1_sharded.json 2> $OUT/bench_rccl_world1_sharded.err
timeout 200 python scripts/mb_kinds.py > $OUT/mb_kinds.txt 2>&1
P=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof -o eager --output-format csv -- python $P/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-extras > $P/$OUT/prof_bench.json 2> $P/$OUT/prof_bench.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/eager_kernel_stats.csv \;
rm -rf $OUT/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/$OUT/pmc_fetch --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_fetch.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/$OUT/pmc_write --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_write.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/$OUT/pmc_mfma --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_mfma.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/$OUT/pmc_mfma128 --output-format csv -- python $P/scripts/pmc_step.py 128 > $P/$OUT/pmc_mfma128.log 2>&1)
EL=$(grep -o "[0-9]* flat elements" $OUT/pmc_fetch.log | grep -o "^[0-9]*")
python scripts/pmc_step_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma ${EL:-153784064} 4 $OUT/gemm_pmc.json > $OUT/pmc_parse.log 2>&1
python scripts/pmc_step_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma128 ${EL:-153784064} 4 $OUT/gemm_pmc_mfma128.json > $OUT/pmc_parse128.log 2>&1
for d in pmc_fetch pmc_write pmc_mfma pmc_mfma128; do find $OUT/$d -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/$d.csv.gz; rm -rf $OUT/$d; done
tail -30 $OUT/pmc_parse.log
