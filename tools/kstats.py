#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: kernel, calls, average / min duration in us.   tools/kstats.py file.csv [min_calls]"""
import csv, sys
mc = int(sys.argv[2]) if len(sys.argv) > 2 else 100
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("ebos::(anonymous namespace)::", "").replace("void ", "")[:78]
    if int(r["Calls"]) >= mc:
        print(f'{n:80s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.2f} us  min {float(r["MinNs"]) / 1e3:8.2f}')
