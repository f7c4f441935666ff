#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 counter_collection.csv:
one row per kernel, one column per counter (summed over launches and over the
counter's instances), plus the number of launches."""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[a-z, ]+>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][-40:]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
names = []
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    agg[k][c] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
print("| kernel | launches | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k in sorted(agg, key=lambda k: -max(agg[k].values())):
    if not k.startswith("k_"):
        continue
    print("| %s | %d | " % (k, len(launches[k])) + " | ".join("%.4g" % agg[k][c] for c in names) + " |")
