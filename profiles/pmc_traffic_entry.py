#!/usr/bin/env python3
"""HBM traffic of one step's path kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), as
MI355X_MICROARCH.md prescribes: separate passes, KB units, gfx950 FETCH_SIZE tallies 64 B per 128-B request
(doubled here).  Per kernel the MEDIAN dispatch (force-only steps) is taken; the path = every kernel of
Pair::compute launched once per step.  Prints the entry bench.py looks up in profiles/pmc_traffic.json.
usage: pmc_traffic_entry.py <dir with pmc_fetch/ pmc_write/> <key> <kernel_source_sha>"""
import collections, csv, glob, json, os, re, sys

root, key, sha = sys.argv[1], sys.argv[2], sys.argv[3]
PATH = ("rebo_centre_kernel", "rebo_centre3_kernel", "rebo_centre_general_kernel", "rebo_lj_tile_kernel", "rebo_lj_gather_kernel", "rebo_gather_kernel",
        "aeam_ptile_kernel", "aeam_tile_density_kernel", "aeam_density_kernel", "aeam_density_ang_kernel", "aeam_embed_kernel",
        "aeam_tile_force_kernel", "aeam_force_kernel", "aeam_force_ang_kernel")


def short(k):
    k = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", k)


def medians(sub, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


(fetch, nf), (write, _) = medians("pmc_fetch", "FETCH_SIZE"), medians("pmc_write", "WRITE_SIZE")
# a force-only step launches every path kernel once; the energy/virial variants of the same kernels run on the
# few thermo steps only and are left out (they would count the pass twice)
most = max([n for k, n in nf.items() if any(k.startswith(p) for p in PATH)] or [0])
per, total = {}, 0.0
for k in sorted(set(fetch) | set(write)):
    if not any(k.startswith(p) for p in PATH) or nf.get(k, 0) * 2 < most:
        continue
    b = (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024.0
    per[k] = {"FETCH_SIZE_KB": fetch.get(k, 0.0), "WRITE_SIZE_KB": write.get(k, 0.0), "hbm_bytes": b}
    total += b
print(json.dumps({key: {"bytes_per_step": total, "kernel_source_sha": sha,
                        "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the path kernels of one force-only step, "
                                "median dispatch per kernel, separate rocprofv3 --pmc passes",
                        "per_kernel": per}}, indent=1))
