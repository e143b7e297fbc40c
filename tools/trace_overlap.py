#!/usr/bin/env python3
"""Do the kernels of two streams overlap?  Reads a rocprofv3 --kernel-trace CSV and prints, for the last N dispatches, name, start and
duration relative to the first of them, plus the total time during which >= 2 kernels were running.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --config 4 ...
    python tools/trace_overlap.py gpurun_out/trace 40 [skip]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # dispatches to leave out at the end
tail = rows[len(rows) - n - skip:len(rows) - skip]
t0 = int(tail[0]["Start_Timestamp"])
ev = []
for r in tail:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ebos::", "").split("<")[0][:44]
    print(f"{nm:46s} start {s/1e3:9.2f} us  dur {(e-s)/1e3:8.2f} us  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')}")
    ev += [(s, 1), (e, -1)]
ev.sort()
depth, last, both, busy = 0, 0, 0, 0
for t, d in ev:
    if depth >= 2: both += t - last
    if depth >= 1: busy += t - last
    depth += d; last = t
print(f"span {ev[-1][0]/1e3:.1f} us, busy {busy/1e3:.1f} us, two or more kernels running {both/1e3:.1f} us")
