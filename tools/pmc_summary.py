#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files.

usage: python tools/pmc_summary.py <dir> [kernel-substring]    (searches <dir> recursively)
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:50]


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_"
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if want not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
        for k in sorted(acc):
            print(f.replace(root, "").split("/")[1], k, {c: "%.4g" % (v / cnt[(k, c)]) for c, v in sorted(acc[k].items())})


if __name__ == "__main__":
    main()
