#!/usr/bin/env python3
"""timeline of one MD step from a rocprofv3 kernel trace: start, duration and the gap before each kernel (us)"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/trace/*/*kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nve_initial" in r["Kernel_Name"]]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
    print("%-46s start %8.1f  dur %7.1f  gap %6.1f" % (n, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    busy += e - s; prev = e
print("step %.1f us, kernels busy %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
