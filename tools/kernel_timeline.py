#!/usr/bin/env python3
"""print the last N kernel launches of a rocprofv3 kernel trace (tools/trace_kernels.sh output) as a timeline"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = glob.glob(d + "/trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e6
    print("%10.3f ms  gap %8.3f  dur %7.3f  %s" % ((s - t0) / 1e6, gap, (e - s) / 1e6, r["Kernel_Name"][:70]))
    prev_end = e
