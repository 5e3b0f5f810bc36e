"""Per-kernel durations and the gaps between consecutive launches from a rocprofv3 kernel trace:
trace_gaps.py <kernel_trace.csv> [last N calls' worth of rows]"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in csv.DictReader(open(sys.argv[1]))))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = rows[-n:]
prev = None
for s, e, k in rows:
    print(f"{(s - prev) / 1e3 if prev else 0:8.1f} us gap  {(e - s) / 1e3:8.1f} us  {k}")
    prev = e
print(f"span {(rows[-1][1] - rows[0][0]) / 1e3:.1f} us")
