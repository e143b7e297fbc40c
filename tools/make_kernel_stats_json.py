#!/usr/bin/env python3
"""profiles/kernel_stats_latest.json from a rocprofv3 --kernel-trace --stats CSV of the default bench command
(tools/collect_round_profiles.sh): average / min / max duration and call count of the path's kernels, stamped with the blob
hash of csrc/iwe_tile_core.h the run was made on -- bench.py quotes it next to its live HIP-event timing and drops it when the
kernel source has changed since.

    python tools/make_kernel_stats_json.py <kernel_stats.csv> <out.json> [commit]"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "event_based_bos_amd", "csrc", "iwe_tile_core.h")
data = open(SRC, "rb").read()
out = {"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (tools/collect_round_profiles.sh)",
       "source_blob_sha": hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest(),
       "commit": sys.argv[3] if len(sys.argv) > 3 else None, "kernels": {}}
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].replace("void ", "").replace("ebos::(anonymous namespace)::", "").split("(")[0]
    if name.startswith(("iwe_", "moments_", "lean_", "bin_", "plan_", "patch_")):
        out["kernels"][name] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 3),
                                "min_us": round(float(r["MinNs"]) / 1e3, 3), "max_us": round(float(r["MaxNs"]) / 1e3, 3)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v["avg_us"] for k, v in out["kernels"].items()}, indent=1))
