"""Per-kernel mean of every counter in rocprofv3 --pmc output directories (counter_collection.csv)."""
import csv
import glob
import os
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")[:40]
            key = (name, row.get("Counter_Name"))
            acc[key][0] += float(row.get("Counter_Value", 0) or 0)
            acc[key][1] += 1
        for (name, c), (v, n) in sorted(acc.items()):
            print("PMC %-40s %-24s mean %.4g over %d dispatches" % (name, c, v / max(1, n), n))
