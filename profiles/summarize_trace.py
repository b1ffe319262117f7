#!/usr/bin/env python3
"""per-kernel summary of a rocprofv3 --kernel-trace csv: calls, median, max, total (ms).
Medians matter: the first/last compute of a bench run tallies energy+virial and is slower."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    d[n[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("%-72s %5s %10s %10s %10s" % ("kernel", "calls", "median_ms", "max_ms", "total_ms"))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-72s %5d %10.4f %10.4f %10.3f" % (k, len(v), v2[len(v2) // 2], v2[-1], sum(v)))
