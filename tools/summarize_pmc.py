"""summarise a rocprofv3 --pmc counter_collection CSV per kernel name (mean per dispatch)"""
import collections, csv, glob, json, sys
root = sys.argv[1]
out = sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        a = acc[name][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
res = {k: {c: {"dispatches": v[0], "mean": v[1] / v[0]} for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k in sorted(res):
    if "fused" in k or "sweep" in k or "gemv" in k:
        print(k, res[k])
