#!/usr/bin/env python3
"""rocprofv3 --pmc csv outputs -> per kernel (name up to the argument list, filtered by a substring), per counter:
median over dispatches.  usage: summarize_pmc2.py <dir> [substring]"""
import collections
import csv
import glob
import json
import os
import re
import sys

root, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        if sub in k:
            res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in res.items()}
for k, c in out.items():
    d = {}
    if "SQ_WAVE_CYCLES" in c and c.get("SQ_WAVE_CYCLES"):
        wc = c["SQ_WAVE_CYCLES"]
        d["valu_issue_frac_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3)
        d["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0) / wc, 3)
        d["wait_inst_any_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    if "SQ_ACTIVE_INST_LDS" in c and c.get("SQ_ACTIVE_INST_LDS"):
        d["lds_conflict_frac_of_lds_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_ACTIVE_INST_LDS"], 3)
    c["derived"] = d
print(json.dumps(out, indent=1))
