"""Developer aid: from a rocprofv3 kernel trace CSV, how much consecutive dispatches of one kernel overlap in time."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
n = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and n % 10 == 0:
        print("dispatch %4d queue %s: start - previous end = %8.1f us, duration %.1f us" % (n, r.get("Queue_Id", "?"), (s - prev_end) / 1e3, (e - s) / 1e3))
    prev_end = e
    n += 1
