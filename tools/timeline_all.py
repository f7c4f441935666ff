import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[len(rows) - n - skip:len(rows) - skip]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
    print("%9.1f  +%8.1f  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id"), r.get("Stream_Id", "?"), m.group(1) if m else r["Kernel_Name"][:30]))
