"""Dev aid: mean per-launch value of every counter per kernel from rocprofv3 --pmc counter_collection.csv files."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for root in sys.argv[1:]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
