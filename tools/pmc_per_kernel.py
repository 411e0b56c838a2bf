"""Per-kernel averages of one counter from a rocprofv3 --pmc run:  python tools/pmc_per_kernel.py DIR COUNTER"""
import csv, glob, os, sys
from collections import defaultdict
d, ctr = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == ctr:
            acc[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    m = sum(v) / len(v); tot += m if len(v) >= 15 else 0
    print(f"{k:62s} n={len(v):4d}  avg {m/1e3:10.1f} k")
print("sum over the kernels launched every tick:", round(tot / 1e6, 2), "M")
