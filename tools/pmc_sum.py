#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter rows per kernel: pmc_sum.py <counter_collection.csv> [substr]"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    print(k, "dispatches", len(cnt[k]), {c: "%.6g" % x for c, x in v.items()})
