"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter."""
import csv, glob, sys, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
