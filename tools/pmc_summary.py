"""Dev aid: per kernel and counter, launches / mean / min / max / sum from rocprofv3 --pmc counter_collection.csv files.
  python tools/pmc_summary.py DIR...            table on stdout
  python tools/pmc_summary.py --csv DIR...      the CSV layout committed under profiles/ (r01f_pmc_*, r01g_pmc_*)"""
import csv, glob, sys, collections
args = sys.argv[1:]
as_csv = "--csv" in args
args = [a for a in args if a != "--csv"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for root in args:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if k.startswith("void "):      # templates: "void ema_k_align_t<32, 8, 4, 0>" -> "ema_k_align_t<32;8;4;0>" (no commas: CSV)
                k = k[5:]
            k = k.replace(", ", ";").replace(",", ";")
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
if as_csv:
    print("kernel,counter,launches,mean_per_launch,min,max,sum_over_run")
for k in sorted(acc):
    if not as_csv:
        print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        if as_csv:
            print(f"{k},{c},{len(v)},{sum(v) / len(v):.6g},{min(v):.6g},{max(v):.6g},{sum(v):.6g}")
        else:
            print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
