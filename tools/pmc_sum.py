#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv per kernel and counter.  usage: pmc_sum.py counter_collection.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
names = sorted({c for v in acc.values() for c in v})
print("%-48s %6s " % ("kernel", "calls") + " ".join("%14s" % c[-14:] for c in names))
for k, v in sorted(acc.items(), key=lambda kv: -max(kv[1].values())):
    calls = max(cnt[(k, c)] for c in names if (k, c) in cnt)
    print("%-48s %6d " % (k, calls) + " ".join("%14.4g" % (v.get(c, 0.0) / calls) for c in names))
