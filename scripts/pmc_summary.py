#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of a kernel."""
import csv, glob, sys
from collections import defaultdict

def summarise(pattern, kernel_substr="loglike_kernel"):
    acc, n = defaultdict(float), defaultdict(int)
    for path in glob.glob(pattern, recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if kernel_substr not in row.get("Kernel_Name", ""):
                    continue
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    return {k: acc[k] / n[k] for k in acc}, {k: n[k] for k in n}

if __name__ == "__main__":
    for pat in sys.argv[1:]:
        mean, cnt = summarise(pat)
        for k in sorted(mean):
            print(f"{k:28s} {mean[k]:18.1f}  (dispatches {cnt[k]})")
