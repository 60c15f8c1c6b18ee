"""Developer aid: aggregate rocprofv3 --pmc counter_collection.csv files per kernel (sum over dispatches)."""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ptrt::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (f, r["Dispatch_Id"]) not in seen:
            seen.add((f, r["Dispatch_Id"]))
for k in sorted(agg):
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print(k)
    for c in sorted(agg[k]): print(f"   {c:36s} {agg[k][c]:.6g}")
