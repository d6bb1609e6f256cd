#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters.
    python tools/pmc_generic.py <rocprof output dir> [name-prefix ...]"""
import collections
import csv
import glob
import re
import sys


def main():
    d = sys.argv[1]
    prefixes = sys.argv[2:] or ["ldati", "conv3d", "events"]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("v2ce::(anonymous namespace)::", "")).replace(" ", "")
            if any(name.startswith(p) for p in prefixes):
                per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(per):
        print(k)
        for c in sorted(per[k]):
            v = per[k][c]
            print(f"    {c:28s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main()
