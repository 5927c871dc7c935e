#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True):
    per = defaultdict(float)
    for row in csv.DictReader(open(f)):
        per[(row["Dispatch_Id"], row["Kernel_Name"].split("(")[0], row["Counter_Name"])] += float(row["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[k][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
