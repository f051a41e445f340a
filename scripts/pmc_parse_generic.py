import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_kernel" in r["Kernel_Name"]:
                agg[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for grid, cs in sorted(agg.items()):
    print("grid", grid, {k: round(sorted(v)[len(v) // 2], 1) for k, v in sorted(cs.items())})
