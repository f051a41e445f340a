#!/bin/bash
# Round 6, closing session at HEAD: PMC passes of the whole step (FETCH_SIZE, WRITE_SIZE, MFMA / busy counters in SEPARATE rocprofv3
# runs, calibrated on cast_kernel) -> gemm_pmc.json stamped with the kernel sources, plus the per-kernel byte table and the calibration
# of the counters in the products' own access pattern; rocprofv3 --kernel-trace --stats of the bench command at every configuration;
# the driver's exact bench command (with the stamped PMC file in place); the full GPU suite; smoke.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_final3
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
grep -E "hbm_read_bytes_per_step|hbm_write_bytes_per_step|kernel_source" $OUT/gemm_pmc.json | head -6
stamp "pmc done"
prof() { local name=$1; shift
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof_$name --output-format csv -- python3 $P/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-others --no-extras --no-preheat "$@" > $P/$OUT/prof_$name.log 2>&1)
  find $OUT/prof_$name -name "*kernel_stats.csv" -exec cp {} $OUT/bench_${name}_kernel_stats.csv \; ; rm -rf $OUT/prof_$name
  head -4 $OUT/bench_${name}_kernel_stats.csv | cut -c1-150; }
prof b4
prof b16 --batch 16
prof b128 --batch 128 --steps 6
prof caption --kind caption
prof pretrain --kind pretrain --batch 6
stamp "rocprof done"
UNIVL_AB=stamps=1 timeout 120 python3 scripts/probe_branches.py > $OUT/probe_stamps_b4.txt 2>&1; tail -12 $OUT/probe_stamps_b4.txt
stamp "stamps done"
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
stamp "driver-form bench rc=$?"
python3 - <<'PY' | tee -a $OUT/summary.txt
import json
j=json.loads([l for l in open("gpurun_out/r06_final3/bench_driver.json") if l.startswith("{")][-1])
print("headline", j["ms_per_step"], j["value"], "preheat", j["preheat"]["block_ms"], j["preheat"]["stable"])
r=j["roofline"]; print("pcie", j["pcie_inclusive"]["ms_per_step"], "family", r.get("family_ms_per_step"), "frac", r.get("frac"), "traffic/step", r.get("traffic_per_step"), r.get("traffic_source"), "alg", r.get("algorithmic_bytes_per_step"))
print("adam", r["adam"]["frac"], "step", r["step"])
print("parity", j.get("parity"))
for o in j.get("other_configs") or []: print(o.get("name"), o.get("ms_per_step"), o.get("value"), o.get("unit"), o.get("error"), o.get("roofline"))
print("cpu", j["cpu_baseline"] and j["cpu_baseline"].get("value"), j["cpu_baseline"] and j["cpu_baseline"].get("cores"), j["cpu_baseline"] and j["cpu_baseline"].get("kind"))
PY
timeout 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-others > $OUT/bench_50_10.json 2>/dev/null
echo "50/10: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_50_10.json | head -1)" | tee -a $OUT/summary.txt
timeout 300 python3 bench.py --force-dp --cfg3-row --no-others --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_force_dp_cfg3_row.json 2> $OUT/bench_force_dp.err
stamp "bench done"
timeout 1700 python3 -m pytest tests/ -q -m gpu --durations=12 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
cp gpurun_out/parity_errors.json $OUT/ 2>/dev/null
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_gpu.log | tail -6
timeout 100 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
stamp "done"
