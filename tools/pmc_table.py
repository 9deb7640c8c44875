"""Per-kernel averages of rocprofv3 --pmc passes, one line per kernel: usage: pmc_table.py folder... (csv output folders).
Counters of all folders are merged by kernel name; FETCH_SIZE / WRITE_SIZE are printed in MB per launch (FETCH_SIZE both
raw and x2: on gfx950 wide coalesced reads are tallied at half their bytes, MI355X_MICROARCH.md section HBM)."""
import csv, glob, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for folder in sys.argv[1:]:
    for path in glob.glob(f"{folder}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            acc[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kernel in sorted(acc):
    parts = []
    for name, vals in sorted(acc[kernel].items()):
        tail = vals[-3:]
        avg = sum(tail) / len(tail)
        if name == "FETCH_SIZE":
            parts.append(f"FETCH_SIZE {avg * 1024 / 1e6:9.2f} MB (x2: {avg * 2048 / 1e6:9.2f})")
        elif name == "WRITE_SIZE":
            parts.append(f"WRITE_SIZE {avg * 1024 / 1e6:9.2f} MB")
        else:
            parts.append(f"{name} {avg:14.0f}")
    print(f"{kernel:48s} launches {len(next(iter(acc[kernel].values()))):3d}  " + "  ".join(parts))
