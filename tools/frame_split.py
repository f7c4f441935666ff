#!/usr/bin/env python3
"""Per-kernel time of the last K frames of a rocprofv3 kernel trace (csv) of
bench.py: launches, total and mean duration per kernel, and the sum against the
frames' span.  A frame ends with its k_resolve (one batch per frame) followed by
k_unpack.   usage: frame_split.py trace.csv K"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[a-z, ]+>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][-40:]


rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_unpack" in r["Kernel_Name"]]
if len(ends) < k + 1:
    sys.exit("trace holds fewer than %d frames" % (k + 1))
lo, hi = ends[-k - 1] + 1, ends[-1] + 1
frames = rows[lo:hi]
span = (int(frames[-1]["End_Timestamp"]) - int(frames[0]["Start_Timestamp"])) / 1e6
agg = collections.OrderedDict()
for r in frames:
    a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("# last %d frames of the trace: %.3f ms per frame from the first kernel's start to the last kernel's end" % (k, span / k))
print()
print("| kernel | launches / frame | ms / frame | mean ms / launch | share |")
print("|---|---|---|---|---|")
tot = sum(a[1] for a in agg.values())
for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %.1f | %.3f | %.4f | %.1f %% |" % (name, n / k, ms / k, ms / n, 100.0 * ms / tot))
print("| sum of kernel durations | | %.3f | | |" % (tot / k))
print("| not covered by a kernel (launch gaps, host round trips) | | %.3f | | |" % ((span - tot) / k))
