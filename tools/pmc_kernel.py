"""Average PMC counter values per kernel from rocprofv3 --pmc csv output folders.  usage: pmc_kernel.py kernel_prefix folder..."""
import csv, glob, sys
from collections import defaultdict
prefix = sys.argv[1]
for folder in sys.argv[2:]:
    acc = defaultdict(list)
    for path in glob.glob(f"{folder}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Kernel_Name"].startswith(prefix):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, vals in sorted(acc.items()):
        tail = vals[-3:]
        print(f"{folder}: {name:28s} launches {len(vals):3d}  last-3 avg {sum(tail)/len(tail):16.0f}")
