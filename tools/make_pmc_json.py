#!/usr/bin/env python3
"""profiles/pmc_latest.json from rocprofv3 --pmc collections of tools/profile_step.py (separate passes, --kernel-trace only):

    python tools/make_pmc_json.py <dir with the passes' output> <out.json> [commit]

Per kernel (keyed by what it does, not by its full template name):
  hbm_bytes_per_launch   2 x FETCH_SIZE + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts
                         128-B requests as 64 B for wide coalesced reads: x2; WRITE_SIZE is exact; both in KiB)
  SQ_INSTS_VALU / SQ_INSTS_LDS / ...   wave-instructions per launch, summed over the chip (bench.py's roofline_issue)
Keys: "<kernel>" for the dense-flow instantiation with a built halo (what BENCH lines have always quoted),
"<kernel><GRID>", "<kernel><UNIFORM>", and "...,DYN>" for the run-time-window variants; the batched accumulate pass is per
LAUNCH of 16 windows ("windows" says so).  Everything is stamped with the blob hash of csrc/iwe_tile_core.h: bench.py drops a
stale collection to null."""
import csv, glob, hashlib, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "event_based_bos_amd", "csrc", "iwe_tile_core.h")


def blob_sha(path):  # = git hash-object (bench.py recomputes it and drops a stale `traffic` to null)
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def key_of(name):
    """kernel template name -> the key bench.py asks for"""
    name = name.replace("void ", "").replace("ebos::(anonymous namespace)::", "")
    base = name.split("<")[0].split("(")[0]
    m = re.search(r"<([^>]*)>", name)
    args = [a.strip() for a in m.group(1).split(",")] if m else []
    flag = lambda i: len(args) > i and args[i] == "true"
    if base == "iwe_slab_accumulate_kernel":      # <TH, TW, HALO, HAS_W, MODE, FMT, UNIFORM, GRID, DYN>
        uni, grid, dyn = flag(6), flag(7), flag(8)
        if len(args) > 5 and args[5] in ("0", "(ebos::(anonymous namespace)::EvFormat)0") and not uni:  # FMT_XY: the general 12 B/event format
            return "iwe_slab_accumulate_kernel<XY" + (",W>" if flag(3) else ">"), name
        if flag(3) and not uni:  # per-event weights on the compact format (10 B/event)
            return "iwe_slab_accumulate_kernel<W>", name
    elif base == "iwe_slab_accumulate_batch_kernel":  # <TH, TW, HALO, GRID, DYN, UNIFORM>
        uni, grid, dyn = flag(5), flag(3), flag(4)
    elif base == "iwe_dense_tiled_bwd_kernel":    # <TH, TW, HALO, HAS_W, FMT, UNIFORM, GRID, DYN>
        uni, grid, dyn = flag(5), flag(6), flag(7)
    elif base in ("iwe_slab_combine4_kernel", "iwe_slab_combine4_batch_kernel"):  # <TH, TW, HALO, DYN>
        uni, grid, dyn = False, False, flag(3)
    elif base == "cmax_resident_kernel":          # <TH, TW, HALO, UNI, FRAC, CONTRAST>: the patch-grid loop, or the 2-DoF one
        tags = (["UNI"] if flag(3) else []) + (["FRAC"] if flag(4) else [])
        contrast = args[5].strip() if len(args) > 5 else "0"
        tags += {"0": [], "1": ["BLUR"], "2": ["GM"]}.get(contrast, [contrast])
        return base + ("<" + ",".join(tags) + ">" if tags else ""), name
    else:
        return base, name
    kind = "UNIFORM" if uni else ("GRID" if grid else "DENSE")
    if kind == "DENSE" and not dyn:
        return base, name
    return f"{base}<{kind}{',DYN' if dyn else ''}>", name


acc = defaultdict(lambda: defaultdict(list))
templates = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k, full = key_of(r["Kernel_Name"])
        if not k.startswith(("iwe_", "moments", "patch_grad", "gradmag", "cmax_resident")):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        templates.setdefault(k, full[:200])
out = {}
for k, v in sorted(acc.items()):
    e = {"template": templates[k]}
    for c, vals in v.items():
        e[c] = sum(vals) / len(vals)
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        e["FETCH_SIZE_bytes_raw"] = e.pop("FETCH_SIZE") * 1024
        e["WRITE_SIZE_bytes"] = e.pop("WRITE_SIZE") * 1024
        e["hbm_bytes_per_launch"] = 2 * e["FETCH_SIZE_bytes_raw"] + e["WRITE_SIZE_bytes"]
    out[k] = e
sizes = {}
try:
    sizes = json.load(open(os.path.join(sys.argv[1], "workloads.json")))
except (OSError, ValueError):
    pass
for k, e in out.items():
    mode = "uniform" if "UNIFORM" in k else ("grid" if "GRID" in k or k.startswith("cmax_resident") else ("general" if "<XY" in k else "dense"))
    if mode in sizes:
        e["events"] = sizes[mode].get("events")
        if "batch" in k:
            e["windows"] = sizes[mode].get("windows_per_launch")
        if k.startswith("cmax_resident"):
            e["iterations_per_launch"] = sizes[mode].get("resident_iterations_per_launch")
res = {"source": "rocprofv3 --pmc (separate passes: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES | "
                 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU) of tools/profile_step.py --mode all; "
                 "FETCH_SIZE x2 (gfx950 coalesced-read correction)",
       "kernels": out,
       "source_blob_sha": blob_sha(SRC),
       "commit": (sys.argv[3] if len(sys.argv) > 3 else None),
       "kernel_template": templates.get("iwe_slab_accumulate_kernel"),
       "iwe_slab_accumulate_hbm_bytes_per_launch": out.get("iwe_slab_accumulate_kernel", {}).get("hbm_bytes_per_launch")}
json.dump(res, open(sys.argv[2], "w"), indent=1)
for k, e in out.items():
    print(f"{k:58s} hbm {e.get('hbm_bytes_per_launch', float('nan')) / 1e6:9.2f} MB  VALU {e.get('SQ_INSTS_VALU', float('nan')):12.0f}  "
          f"LDS {e.get('SQ_INSTS_LDS', float('nan')):11.0f}  events {e.get('events')}")
