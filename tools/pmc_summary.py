"""Summarise rocprofv3 counter_collection CSVs per kernel name (sum over dispatches / dispatch count)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            if "conv_gemm" not in k:
                continue
            k = k.split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        n = cnt[k][c]
        print(f"   {c:36s} {agg[k][c] / n:16.1f}  (avg of {n} dispatches)")
