"""python scripts/pmc_gemm_calib_parse.py <fetch dir> <write dir>: bytes per FETCH_SIZE / WRITE_SIZE unit of the GEMM family's access
patterns against cast_kernel's (scripts/pmc_gemm_calib.py)."""
import csv
import glob
import json
import sys

n_el = 64 << 20
H, I, R = 768, 3072, 24


def rows(d, counter):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(out)


def main(fd, wd, out=None):
    med = lambda v: sorted(v)[len(v) // 2]
    res = {}
    for which, d, counter in (("read", fd, "FETCH_SIZE"), ("write", wd, "WRITE_SIZE")):
        rs = rows(d, counter)
        cast = [v for _, k, v in rs if "cast_kernel" in k]
        unit = (4.0 if which == "read" else 2.0) * n_el / med(cast)
        gem = [v for _, k, v in rs if "gemm_kernel" in k]          # launch order: fwd1 x R, fwd3 x R, dgrad1 x R, dgrad3 x R, wgrad x R
        assert len(gem) == 5 * R, len(gem)
        groups = dict(fwd1=gem[:R], fwd3=gem[R:2 * R], dgrad1=gem[2 * R:3 * R], dgrad3=gem[3 * R:4 * R], wgrad=gem[4 * R:])
        known = dict(read=dict(fwd1=I * H * 2 + 64 * H * 2, fwd3=I * H * 2 + 192 * H * 2, dgrad1=I * H * 2 + 64 * H * 2, dgrad3=I * H * 2 + 192 * H * 2,
                               wgrad=192 * (H + I) * 2),
                     write=dict(fwd1=64 * I * 2, fwd3=192 * I * 2, dgrad1=64 * I * 2, dgrad3=192 * I * 2, wgrad=H * I * 4))[which]
        res[which] = dict(cast_bytes_per_unit=unit)
        for g, v in groups.items():
            cal = med(v) * unit                                    # what the cast calibration says this launch moved
            res[which][g] = dict(known_bytes=known[g], bytes_by_cast_calibration=round(cal), ratio=round(cal / known[g], 3))
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
