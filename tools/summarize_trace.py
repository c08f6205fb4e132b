#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --kernel-trace output directory, template arguments kept (k_update<geom, prior, maxv, u8>)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
files = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)
agg = defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name).replace("pm::", "")
        agg[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':44s} calls   total ms     avg ms     min ms     max ms    share")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:44]:44s} {len(v):5d} {sum(v):10.3f} {sum(v) / len(v):10.4f} {min(v):10.4f} {max(v):10.4f}  {100 * sum(v) / tot:5.1f}%")
