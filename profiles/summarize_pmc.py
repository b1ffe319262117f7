#!/usr/bin/env python3
"""summarize rocprofv3 --pmc csv outputs: per kernel, per counter, mean over dispatches (force-only steps)"""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")) + glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = ("lj_tile" if "lj_tile" in k else "lj_cubic" if "lj_cubic" in k else "lj_gather" if "lj_gather" in k else "centre16" if "centre_kernel<16" in k else "centre12" if "centre_kernel<12" in k
                 else "centre8_overflow_pass" if "centre_kernel<8, true" in k else "centre8" if "centre_kernel<8" in k else "centre_general" if "centre_general" in k else "centre4" if "centre_kernel<4" in k else "centre3" if "centre3_kernel" in k
                 else "aeam_ptile" if "aeam_ptile" in k else "aeam_tile_force" if "aeam_tile_force" in k else "aeam_tile_density" if "aeam_tile_density" in k else "aeam_force" if "aeam_force_kernel" in k else "aeam_density" if "aeam_density_kernel" in k else None)
        if short:
            res[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in res.items():
    out[k] = {}
    for c, v in cs.items():
        v = sorted(v)
        # drop the two eflag/vflag dispatches (first/last) by taking the median
        out[k][c] = v[len(v) // 2]
print(json.dumps(out, indent=1))
