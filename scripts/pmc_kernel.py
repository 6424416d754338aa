"""Mean of every counter of a rocprofv3 --pmc csv for the kernels whose name contains argv[2]."""
import csv, sys, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if sys.argv[2] in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
for c, v in sorted(acc.items()):
    v = v[2:] if len(v) > 4 else v
    print(f"{c:32s} {sum(v) / len(v):16.1f}  ({len(v)} dispatches)")
