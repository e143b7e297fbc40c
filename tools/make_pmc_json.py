#!/usr/bin/env python3
"""profiles/pmc_latest.json from a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collection (tools/profile_step.py):
HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads: x2; WRITE_SIZE is exact; both in KiB)."""
import csv, glob, hashlib, json, os, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "event_based_bos_amd", "csrc", "iwe_tiled.hip")


def blob_sha(path):  # = git hash-object (bench.py recomputes it and drops a stale `traffic` to null)
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


acc = defaultdict(lambda: defaultdict(list))
templates = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            name = r["Kernel_Name"].replace("void ", "").replace("ebos::(anonymous namespace)::", "").split("<")[0].split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            templates.setdefault(name, r["Kernel_Name"].replace("ebos::(anonymous namespace)::", "")[:160])
out = {}
for k, v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v and k.startswith(("iwe_", "moments")):
        fetch = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 1024
        write = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024
        out[k] = {"FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write}
res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/profile_step.py, 10M events 1280x720, "
                 "compact plan, tile 45x80 halo 32; FETCH_SIZE x2 (gfx950 coalesced-read correction)",
       "kernels": out,
       "source_blob_sha": blob_sha(SRC),
       "commit": (sys.argv[3] if len(sys.argv) > 3 else None),
       "kernel_template": templates.get("iwe_slab_accumulate_kernel"),
       "iwe_slab_accumulate_hbm_bytes_per_launch": out.get("iwe_slab_accumulate_kernel", {}).get("hbm_bytes_per_launch")}
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
