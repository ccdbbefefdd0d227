"""Sums rocprofv3 counter_collection CSVs per kernel (mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            short = name.split("(")[0].replace("void ", "").replace("dnagpu::", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for kern in sorted(acc):
    if any(x in kern for x in ("leaves", "scatter", "level_hist", "fb_", "sk_", "mini")):
        print(kern)
        for c in sorted(acc[kern]):
            v = acc[kern][c]
            print(f"    {c:28s} mean/dispatch {sum(v)/len(v):16.1f}   (n={len(v)})")
