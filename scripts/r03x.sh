#!/bin/bash
# Round 3, closing session at HEAD (half-width tiles by slot fill, 4-wave deep grouped weight gradients): L2 counters of a layer's
# grouped weight gradients (isolated), data-parallel step at 128 pairs with the group on 8 / 4 waves, the three PMC passes of the
# step at 4 pairs (-> profiles/r03_gemm_pmc.json, stamped with the kernel sources), bench lines, GEMM + 128-pair tests, smoke.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
BUDGET=${BUDGET:-200}
OUT=gpurun_out/r03x
mkdir -p $OUT
P=$PWD
left() { echo $(( BUDGET - ( $(date +%s) - T0 ) )); }
lim() { local want=$1 l; l=$(left); if [ $l -lt 5 ]; then echo 0; elif [ $l -lt $want ]; then echo $l; else echo $want; fi; }
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
# 1. stamp first: the three PMC passes of the step at 4 pairs
t=$(lim 40); [ $t -gt 15 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/$OUT/pmc_fetch --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_fetch.log 2>&1)
t=$(lim 40); [ $t -gt 15 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/$OUT/pmc_write --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_write.log 2>&1)
t=$(lim 40); [ $t -gt 15 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/$OUT/pmc_mfma --output-format csv -- python $P/scripts/pmc_step.py > $P/$OUT/pmc_mfma.log 2>&1)
EL=$(grep -o "[0-9]* flat elements" $OUT/pmc_fetch.log | grep -o "^[0-9]*")
python scripts/pmc_step_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma ${EL:-153784064} 4 $OUT/gemm_pmc.json > $OUT/pmc_parse.log 2>&1 && cp $OUT/gemm_pmc.json profiles/r03_gemm_pmc.json
for k in fetch write mfma; do find $OUT/pmc_$k -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_$k.csv.gz; rm -rf $OUT/pmc_$k; done
stamp "pmc done"
# 2. the bench lines
t=$(lim 90); [ $t -gt 30 ] && { timeout $t python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json; grep -o '"traffic": [0-9.a-z]*' $OUT/bench.json; }
stamp "bench done"
t=$(lim 30); [ $t -gt 12 ] && { timeout $t python bench.py --batch 128 --steps 60 --warmup 8 --no-cpu-baseline --no-extras > $OUT/bench_b128.json 2> $OUT/bench_b128.err; echo "b128 $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128.json)"; }
t=$(lim 30); [ $t -gt 12 ] && { timeout $t python bench.py --batch 16 --steps 150 --warmup 15 --no-cpu-baseline --no-extras > $OUT/bench_b16.json 2> $OUT/bench_b16.err; echo "b16 $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b16.json)"; }
stamp "bench lines done"
# 3. tests on the new defaults
t=$(lim 80); [ $t -gt 20 ] && timeout $t python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "gemm or b128" -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
t=$(lim 40); [ $t -gt 10 ] && timeout $t python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
stamp "tests done"
# 4. data-parallel step at 128 pairs (RCCL world 1): the deep grouped weight gradients on 8 / 4 waves
for v in "UNIVL_GEMM_NC64_WAVES=8" "UNIVL_GEMM_NC64_WAVES=4"; do
  t=$(lim 30); [ $t -gt 12 ] || break
  env $v timeout $t python bench.py --batch 128 --steps 40 --warmup 8 --force-dp --no-cpu-baseline --no-extras > $OUT/bench_b128_dp_$v.json 2> $OUT/bench_b128_dp_$v.err
  echo "dp b128 $v $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_b128_dp_$v.json)" | tee -a $OUT/ab_dp_b128.txt
done
stamp "dp ab done"
# 5. L2 counters of the isolated grouped weight gradients
t=$(lim 40); [ $t -gt 15 ] && (cd /tmp && timeout $t rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $P/$OUT/pmc_l2 --output-format csv -- python $P/scripts/mb_gemm_variants.py --rows "" --group-rows 6144 --variants 64/2/4,128/2/8,256/3/8 --out $P/$OUT/mb_group_pmc.json > $P/$OUT/pmc_l2.log 2>&1)
python - <<'PY' > $OUT/pmc_l2_summary.txt 2>&1
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.Counter()); n = collections.Counter()
for f in glob.glob("gpurun_out/r03x/pmc_l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_group" not in k: continue
        k = k[k.find("gemm_group"):][:90]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, c in tot.items():
    l = max(n[(k, x)] for x in c)
    print(k, "launches", l, {x: "%.3e" % (v / l) for x, v in c.items()}, "hit rate %.3f" % (c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
PY
cat $OUT/pmc_l2_summary.txt | cut -c1-400
find $OUT/pmc_l2 -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_l2.csv.gz; rm -rf $OUT/pmc_l2
stamp "end"
