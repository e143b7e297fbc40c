#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel (per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").replace("ebos::(anonymous namespace)::", "").split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(t in k for t in ("iwe_", "moments", "variance", "gradmag")):
        continue
    print(k[:90])
    for c, v in sorted(acc[k].items()):
        print(f"    {c:34s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
