"""Turn the two rocprofv3 counter CSVs (FETCH_SIZE pass, WRITE_SIZE pass) of scripts/pmc_adam.py into
profiles/r01_adam_pmc.json.  Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of the fabric request counters and, on gfx950, wide coalesced
reads are tallied at half their size -- so both counters are CALIBRATED on a kernel with exactly known traffic in the
same access-pattern class (univl_cast_bf16: 4 B read + 2 B written per element) and the adam kernel's counts are
scaled by the calibration factors."""
import csv
import glob
import json
import sys


def per_kernel(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        out.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return out


def pick(d, key):
    for k, v in d.items():
        if key in k:
            return v
    raise SystemExit("kernel %s not found in %s" % (key, list(d)[:10]))


def main(fetch_dir, write_dir, elements, params, out):
    f = per_kernel(glob.glob(fetch_dir + "/**/*counter_collection.csv", recursive=True)[0], "FETCH_SIZE")
    w = per_kernel(glob.glob(write_dir + "/**/*counter_collection.csv", recursive=True)[0], "WRITE_SIZE")
    med = lambda v: sorted(v)[len(v) // 2]
    cast_f, cast_w = med(pick(f, "cast_kernel")[-4:]), med(pick(w, "cast_kernel")[-4:])
    adam_f, adam_w = med(pick(f, "adam_apply")[-6:]), med(pick(w, "adam_apply")[-6:])
    kf = 4.0 * elements / cast_f          # true bytes per counter unit, reads
    kw = 2.0 * elements / cast_w          # writes
    rd, wr = adam_f * kf, adam_w * kw
    alg_r, alg_w = 16.0 * params, 14.0 * params      # p,g,m,v read; p,m,v + bf16 shadow written
    res = dict(kernel="adam_apply_kernel", counters=dict(FETCH_SIZE=adam_f, WRITE_SIZE=adam_w),
               calibration=dict(kernel="cast_kernel (4 B read + 2 B written per element, %d elements)" % elements,
                                FETCH_SIZE=cast_f, WRITE_SIZE=cast_w, read_bytes_per_unit=kf, write_bytes_per_unit=kw),
               hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
               algorithmic_bytes_per_launch=alg_r + alg_w, traffic_over_algorithmic=(rd + wr) / (alg_r + alg_w))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
