"""Summarise rocprofv3 --pmc CSV output: per kernel, mean of every counter over its dispatches."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("saf::", "")
            if "afstft" in k or "eq_kernel" in k or "gemm" in k or "pconv" in k or "binaural" in k or "cov" in k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
