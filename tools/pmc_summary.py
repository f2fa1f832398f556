#!/usr/bin/env python
"""Per-kernel averages of every counter in a rocprofv3 counter_collection.csv (any number of files).
usage: python tools/pmc_summary.py <csv> [<csv> ...]"""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("stm::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "stm_k_" not in k: continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
