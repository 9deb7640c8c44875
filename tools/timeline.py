"""Kernel timeline of the last full step from a rocprofv3 --kernel-trace CSV:
    python tools/timeline.py <kernel_trace.csv> [marker kernel-name prefix ...]
Prints every launch of the last step (start relative to the marker kernel's start, duration, gap to the previous
launch).  A step begins at a launch of the marker (default: the Q1 scan kernel)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
markers = tuple(sys.argv[2:]) or ("k_agg_jit", "void k_agg_main")
scan = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(markers)]
if len(scan) < 3:
    raise SystemExit("need at least three steps in the trace")
a, b = scan[-2], scan[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {r['Kernel_Name'][:60]}")
    prev_end = e
print(f"step period {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
