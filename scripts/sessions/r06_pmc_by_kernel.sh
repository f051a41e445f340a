#!/bin/bash
# Round 6: HBM read / write bytes of one training step PER KERNEL (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes of
# scripts/pmc_step.py, the optimizer update as launches of its own) -> which launches own the family's 1.58 GB of reads.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_pmc
mkdir -p $OUT
P=$PWD
for k in fetch write; do
  c=$( [ $k = fetch ] && echo FETCH_SIZE || echo WRITE_SIZE )
  (cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $c -d $P/$OUT/pmc_$k --output-format csv -- python3 $P/scripts/pmc_step.py > $P/$OUT/pmc_$k.log 2>&1)
done
python3 scripts/pmc_parse_by_kernel.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/by_kernel.txt 2>&1
python3 - <<'PY' > gpurun_out/r06_pmc/by_kernel_totals.txt
import collections, csv, glob, re
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for which, idx in (("fetch", 0), ("write", 1)):
    for f in glob.glob("gpurun_out/r06_pmc/pmc_%s/**/*counter_collection.csv" % which, recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)[-90:]
            tot[name][idx] += float(r["Counter_Value"])
            if idx == 0:
                tot[name][2] += 1
# FETCH_SIZE on gfx950 reports 1/2 of wide streaming reads (guide): calibrate both on cast_kernel (4 B read + 2 B written per element)
n_el = 153784064
cast = [v for k, v in tot.items() if "cast_kernel" in k][0]
fr, fw = 4.0 * n_el * (cast[2]) / cast[0], 2.0 * n_el * (cast[2]) / cast[1]
print("calibration on cast_kernel: %.1f bytes per FETCH unit, %.1f per WRITE unit; 4 steps" % (fr, fw))
rows = sorted(tot.items(), key=lambda kv: -(kv[1][0] * fr + kv[1][1] * fw))
for name, (a, b, n) in rows[:28]:
    print("%-92s launches/step %6.1f  read %8.1f MB/step  write %8.1f MB/step" % (name, n / 4.0, a * fr / 4 / 1e6, b * fw / 4 / 1e6))
PY
for k in fetch write; do find $OUT/pmc_$k -name "*counter_collection.csv" -exec gzip -c {} \; > $OUT/pmc_$k.csv.gz; rm -rf $OUT/pmc_$k; done
cat $OUT/by_kernel_totals.txt
