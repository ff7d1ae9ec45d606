"""Sums rocprofv3 --pmc CSV output per kernel name: python scripts/pmc_summary.py DIR [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and pat not in k:
            continue
        acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k[:60]].add(r["Dispatch_Id"])
    print(os.path.relpath(f, d))
    for k, c in acc.items():
        print("  %s  (%d dispatches)" % (k, len(n[k])))
        for name, v in sorted(c.items()):
            print("    %-28s %.6g  (per dispatch %.6g)" % (name, v, v / len(n[k])))
