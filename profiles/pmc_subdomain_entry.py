#!/usr/bin/env python3
"""HBM traffic of ONE step of rank 0's sub-domain in an N-brick run, from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE; separate runs, KB units, gfx950 FETCH_SIZE counts 64 B per 128-B request and is doubled -- as
MI355X_MICROARCH.md prescribes) over profiles/subdomain_step.py, whose last `steps` steps are rank 0 stepping alone:
per path kernel the last steps x (launches per step) dispatches are summed and divided by the steps.
usage: pmc_subdomain_entry.py <dir with pmc_fetch/ pmc_write/> <key e.g. rebomos:24x24x24:8> <kernel_source_sha> <steps>"""
import collections, csv, glob, json, os, re, sys

root, key, sha, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
AEAM = key.startswith("aeam")
PATH = ("aeam_ptile_kernel", "aeam_tile_density_kernel", "aeam_density_ang_kernel", "aeam_embed_kernel",
        "aeam_tile_force_kernel", "aeam_force_ang_kernel", "aeam_density_kernel", "aeam_force_kernel") if AEAM else \
       ("rebo_centre_kernel", "rebo_centre3_kernel", "rebo_centre_general_kernel", "rebo_lj_tile_kernel", "rebo_lj_gather_kernel",
        "rebo_gather_kernel")


def short(k):
    k = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", k)


def per_step(sub, counter):
    """sum per kernel over the window of the last `steps` steps (delimited by the force-only Lennard-Jones dispatches,
    one per step), divided by the steps; also the launches per step seen in that window"""
    rows = []
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
    rows.sort()
    # one dispatch per step that closes the path: the force-only Lennard-Jones kernel / aeam's last force-tile launch
    if AEAM:
        emb = [d for d, k, _ in rows if k.startswith("aeam_embed_kernel")]
        frc = [d for d, k, _ in rows if k.startswith("aeam_tile_force_kernel")]
        import bisect
        lj = []
        for e, nxt in zip(emb, emb[1:] + [1 << 62]):
            i = bisect.bisect_left(frc, nxt) - 1
            if i >= 0 and frc[i] > e:
                lj.append(frc[i])
    else:
        lj = [d for d, k, _ in rows if k.startswith("rebo_lj_tile") and "<false" in k]
    if len(lj) <= steps:
        return {}
    start = lj[-steps - 1]                      # everything after the Lennard-Jones kernel of the step before the window
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for d, k, v in rows:
        if d > start and any(k.startswith(p) for p in PATH):
            tot[k] += v
            cnt[k] += 1
    return {k: (tot[k] / steps, cnt[k] / steps) for k in tot}


fetch, write = per_step("pmc_fetch", "FETCH_SIZE"), per_step("pmc_write", "WRITE_SIZE")
per, total = {}, 0.0
for k in sorted(set(fetch) | set(write)):
    fkb, lps = fetch.get(k, (0.0, 1))
    wkb = write.get(k, (0.0, 1))[0]
    b = (2.0 * fkb + wkb) * 1024.0
    per[k] = {"FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb, "hbm_bytes": b, "launches_per_step": lps}
    total += b
print(json.dumps({key: {"bytes_per_step": total, "kernel_source_sha": sha,
                        "note": "rank 0's sub-domain of the N-brick run, all bricks set up on one GPU, rank 0 stepping alone "
                                "(profiles/subdomain_step.py): (2*FETCH_SIZE + WRITE_SIZE)*1024 of the path kernels summed "
                                "over the last steps and divided by them, separate rocprofv3 --pmc passes",
                        "per_kernel": per}}, indent=1))
