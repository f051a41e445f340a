#!/bin/bash
# Round 6, last session: after the default went back to libm's erff in gemm_tile's GELU epilogues (UNIVL_GELU_FAST=0) -- the golden cases in
# both modes + the operand-pairs tests (parity_errors.json), the PMC passes (stamp for the final kernel sources), the driver's command,
# the full GPU suite, smoke.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_final6
mkdir -p $OUT
P=$PWD
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
for k in fetch write; do
  c=$( [ $k = fetch ] && echo FETCH_SIZE || echo WRITE_SIZE )
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $c -d $P/$OUT/pmc_$k --output-format csv -- python3 $P/scripts/pmc_step.py > $P/$OUT/pmc_$k.log 2>&1)
done
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/$OUT/pmc_mfma --output-format csv -- python3 $P/scripts/pmc_step.py > $P/$OUT/pmc_mfma.log 2>&1)
EL=$(grep -o "[0-9]* flat elements" $OUT/pmc_fetch.log | grep -o "^[0-9]*")
python3 scripts/pmc_step_parse.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma ${EL:-153784064} 4 $OUT/gemm_pmc.json > $OUT/pmc_parse.log 2>&1
for k in fetch write mfma; do find $OUT/pmc_$k -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_$k.csv.gz; rm -rf $OUT/pmc_$k; done
cp $OUT/gemm_pmc.json profiles/r06_gemm_pmc.json
stamp "pmc done"
(cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof_b4 --output-format csv -- python3 $P/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-others --no-extras --no-preheat > $P/$OUT/prof_b4.log 2>&1)
find $OUT/prof_b4 -name "*kernel_stats.csv" -exec cp {} $OUT/bench_b4_kernel_stats.csv \; ; rm -rf $OUT/prof_b4
stamp "rocprof done"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
stamp "driver-form bench rc=$?"
python3 - <<'PY' | tee -a $OUT/summary.txt
import json
j=json.loads([l for l in open("gpurun_out/r06_final6/bench_driver.json") if l.startswith("{")][-1])
print("headline", j["ms_per_step"], j["value"], "preheat", j["preheat"]["block_ms"], j["preheat"]["stable"])
r=j["roofline"]; print("pcie", j["pcie_inclusive"]["ms_per_step"], "family", r.get("family_ms_per_step"), "frac", r.get("frac"), "traffic/step", r.get("traffic_per_step"), r.get("traffic_source"))
print("adam", r["adam"]["frac"], "step", r["step"])
print("parity", {k: j["parity"].get(k) for k in ("case", "passed", "gglobal", "gmedian", "gate")})
print("like_for_like", j.get("like_for_like", {}).get("change"))
for o in j.get("other_configs") or []: print(o.get("name"), o.get("ms_per_step"), o.get("value"), o.get("unit"), o.get("error"), o.get("skipped"))
print("cpu", j["cpu_baseline"] and j["cpu_baseline"].get("value"), j["cpu_baseline"] and j["cpu_baseline"].get("kind"))
PY
timeout 1700 python3 -m pytest tests/ -q -m gpu --durations=8 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_gpu.log | tail -6
timeout 100 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
stamp "done"
