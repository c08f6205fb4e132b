#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory: per-kernel time from the
kernel trace and per-kernel averages of every PMC counter collected."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]

# header: which build and which command these numbers belong to (bench.py only quotes a profile whose hash equals the tree's)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import bench
    print(f"# kernel_build_sha256: {bench.kernel_build_sha256()}")
except Exception as e:  # noqa: BLE001
    print(f"# kernel_build_sha256: unavailable ({e})")
for key, path in (("command", os.path.join(out, "command.txt")), ("git_head", os.path.join(ROOT, "build", "git_head.txt")),
                  ("k_update_passes_per_dispatch", os.path.join(out, "passes_per_dispatch.txt"))):
    if os.path.exists(path):
        print(f"# {key}: {open(path).read().strip()}")


BY_VARIANT = "--by-variant" in sys.argv   # keep the template arguments of k_update / k_init (one row per instantiation)


def short(name):
    if BY_VARIANT and ("k_update" in name or "k_init" in name):
        import re
        m = re.search(r"(k_update|k_init)<([^>]*)>", name)
        if m:
            return m.group(1) + "<" + m.group(2).replace(" ", "") + ">"
    for key in ("k_update", "k_init", "k_filter", "k_depth_normal", "k_pad", "k_export", "k_eval", "k_prior_raster", "k_prior"):
        if key in name:
            return key
    return name[:40]


files = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if files:
    agg = defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    print("== kernel trace (ms) ==")
    tot = sum(sum(v) for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:18s} calls {len(v):4d}  total {sum(v):9.3f}  avg {sum(v) / len(v):8.4f}  min {min(v):8.4f}  max {max(v):8.4f}  {100 * sum(v) / tot:5.1f}%")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(f"== {os.path.basename(d)}: no counter csv ==")
        continue
    agg = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("Scratch_Size"), row.get("LDS_Block_Size"))
    print(f"== {os.path.basename(d)} (per-dispatch average) ==")
    for k in agg:
        if not (k.startswith("k_update") or k.startswith("k_init")):
            continue
        print(f"  {k}  vgpr/agpr/sgpr/scratch/lds = {meta[k]}")
        for c, v in agg[k].items():
            print(f"      {c:36s} n={len(v):3d} avg={sum(v) / len(v):.6g}")
