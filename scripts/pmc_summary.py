#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel.

    python scripts/pmc_summary.py "<glob of counter_collection.csv>" [kernel-name substring ...]
Default substrings: the log-L kernels, the walk, the prior kernels, the scalar-call server, the FIP kernels."""
import csv, glob, re, sys
from collections import defaultdict

DEFAULT = ("loglike", "slice_walk", "prior_", "scalar_server", "fip_")


def short(name):
    name = re.sub(r"rvll::\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def summarise(pattern, substrs=DEFAULT):
    acc, n = defaultdict(float), defaultdict(int)
    for path in glob.glob(pattern, recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                kn = row.get("Kernel_Name", "")
                if not any(s in kn for s in substrs):
                    continue
                key = (short(kn), row["Counter_Name"])
                acc[key] += float(row["Counter_Value"]); n[key] += 1
    return {k: acc[k] / n[k] for k in acc}, dict(n)


if __name__ == "__main__":
    pats = [a for a in sys.argv[1:] if "*" in a or a.endswith(".csv")]
    subs = tuple(a for a in sys.argv[1:] if a not in pats) or DEFAULT
    for pat in pats:
        mean, cnt = summarise(pat, subs)
        for kern in sorted({k for k, _ in mean}):
            print(f"== {kern}")
            for (k, c) in sorted(mean):
                if k == kern:
                    print(f"{c:28s} {mean[(k, c)]:18.1f}  (dispatches {cnt[(k, c)]})")
