"""Kernel timeline of the LAST run of a rocprofv3 --kernel-trace CSV: python tools/trace_tail.py <kernel_trace.csv> [n]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  + {(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:7.1f}  {r['Kernel_Name'][:70]}")
    prev = e
