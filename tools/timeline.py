"""Kernel timeline of the last frame in a rocprofv3 kernel trace (csv): start
offset, duration and the gap to the previous kernel's end, per kernel."""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_resolve" in r["Kernel_Name"]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
frame = rows[lo:ends[-1] + 1]
t0 = int(frame[0]["Start_Timestamp"])
busy_until = t0
idle = 0
for r in frame:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    import re as _re
    _m = _re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
    name = _m.group(1) if _m else r["Kernel_Name"].split("(")[0][-30:]
    gap = s - busy_until
    if gap > 0:
        idle += gap
    print("%9.1f us  +%8.1f us  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, r.get("Queue_Id", "?"), name[:40]))
    busy_until = max(busy_until, e)
print("frame %.1f us, no kernel running for %.1f us" % ((busy_until - t0) / 1e3, idle / 1e3))
