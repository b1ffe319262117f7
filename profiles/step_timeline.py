#!/usr/bin/env python3
"""timeline of one MD step from a rocprofv3 kernel trace (start, duration and the gap before each kernel, us) and the
mean over the last steps of the trace.  profiles/subdomain_step.py ends with rank 0 stepping alone, so the last steps
are that loop.  usage: step_timeline.py <dir with trace/> [steps to average, default 30] [which step to print, from the end]
(bench.py ends with 20 steps that carry timing events between the kernels and one forced reneighboring: pass e.g.
`60 45` to print a step of the timed region and `60`... the mean then still spans those 20 steps)"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/trace/*/*kernel_trace.csv"))[-1]
navg = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nve_advance" in r["Kernel_Name"] or "nve_initial" in r["Kernel_Name"]]
short = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 10
a, b = idx[-back - 1], idx[-back]    # one step of the final loop, `back` (default ten) from the end
t0 = int(rows[a]["Start_Timestamp"]); prev = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-46s start %8.1f  dur %7.1f  gap %6.1f" % (short(r), (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    busy += e - s; prev = e
print("step %.1f us, kernels busy %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3, b - a))
# means over the last `navg` steps
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # leave the last `skip` steps out of the mean
first, last = idx[-navg - 1 - skip], idx[-1 - skip]
span = (int(rows[last]["Start_Timestamp"]) - int(rows[first]["Start_Timestamp"])) / 1e3 / navg
tot, per = 0.0, {}
for r in rows[first:last]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    per[short(r)] = per.get(short(r), 0.0) + d
print("mean of the last %d steps: step %.1f us, kernels busy %.1f us" % (navg, span, tot / navg))
for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
    print("   %-46s %7.1f us per step" % (k, v / navg))
