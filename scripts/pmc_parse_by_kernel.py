"""Medians of every counter per kernel name (rocprofv3 --pmc output directories as arguments; optional --match substring)."""
import collections, csv, glob, re, sys
args = [a for a in sys.argv[1:] if not a.startswith("--match=")]
match = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--match=")]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if match and not any(m in name for m in match):
                continue
            short = re.sub(r"\(anonymous namespace\)::", "", name)
            short = re.sub(r"\((?:GemmArgs|GroupArgs)[^)]*\)$", "", short)[-100:]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    med = {k: sorted(v)[len(v) // 2] for k, v in sorted(cs.items())}
    n = max(len(v) for v in cs.values())
    print(name, "launches", n)
    print("   ", {k: round(v, 1) for k, v in med.items()})
    wc = med.get("SQ_WAVE_CYCLES")
    if wc:
        extra = {k: round(med[k] / wc, 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if k in med}
        if "SQ_LDS_BANK_CONFLICT" in med and med.get("SQ_LDS_IDX_ACTIVE"):
            extra["lds_conflict_frac"] = round(med["SQ_LDS_BANK_CONFLICT"] / med["SQ_LDS_IDX_ACTIVE"], 3)
        print("    fractions of wave cycles:", extra)
