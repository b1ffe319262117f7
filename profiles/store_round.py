#!/usr/bin/env python3
"""copy a round's collection (gpurun_out/<round>_*, made by profiles/collect_round.sh on the GPU box) into profiles/
and refresh the entries of profiles/pmc_traffic.json that bench.py reports as `traffic`.
usage: python3 profiles/store_round.py r04"""
import json
import os
import shutil
import sys

RND = sys.argv[1] if len(sys.argv) > 1 else "r04"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tab = json.load(open(os.path.join(P, "pmc_traffic.json")))
for tag, main in ((RND + "_rebomos4m", True), (RND + "_aeam1m", True), (RND + "_rebomos4m_300K", False),
                  (RND + "_aeam16m", True)):  # (the last one: profiles/collect.sh by hand, configuration #5 on one GPU)
    d = os.path.join(G, tag)
    if not os.path.isdir(d):
        if not tag.endswith("16m"):
            print("missing", d)
        continue
    # (bench.json of the collection is the line BEFORE the PMC pass: the lines re-collected afterwards -- collect_rNN.sh lines --
    #  are copied over it by hand so that the stored JSON carries `traffic`)
    for src, dst in (("bench.json", "bench.json"), ("kernel_summary.txt", "kernel_summary.txt"),
                     ("rocprofv3_kernel_stats.csv", "rocprofv3_kernel_stats.csv"), ("pmc_entry.json", "pmc_fetch_write.json")):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(P, f"{tag}_{dst}"))
    if main and os.path.exists(os.path.join(d, "pmc_entry.json")):
        ent = json.load(open(os.path.join(d, "pmc_entry.json")))
        for k, v in ent.items():
            if v.get("bytes_per_step", 0) > 0:
                tab[k] = v
                print("pmc entry", k, v["bytes_per_step"], v["kernel_source_sha"])
for n in (2, 4, 8):
    sd = os.path.join(G, RND + "_subdomain%d" % n)
    if os.path.exists(os.path.join(sd, "timeline.txt")):
        with open(os.path.join(P, RND + "_subdomain%d_step_timeline.txt" % n), "w") as f:
            f.write(open(os.path.join(sd, "subdomain.json")).read() + "\n" + open(os.path.join(sd, "timeline.txt")).read())
    pe = os.path.join(sd, "pmc_entry.json")
    if os.path.exists(pe) and os.path.getsize(pe) > 10:
        ent = json.load(open(pe))
        for k, v in ent.items():
            if v.get("bytes_per_step", 0) > 0:
                tab[k] = v
                print("pmc entry", k, v["bytes_per_step"], v["kernel_source_sha"])
# aeam sub-domains (BASELINE.json configs[4], profiles/collect_subdomain_aeam.sh)
for n in (2, 4, 8):
    sd = os.path.join(G, RND + "_aeam_subdomain%d" % n)
    if os.path.exists(os.path.join(sd, "timeline.txt")):
        with open(os.path.join(P, RND + "_aeam_subdomain%d_step_timeline.txt" % n), "w") as f:
            f.write(open(os.path.join(sd, "subdomain.json")).read() + "\n" + open(os.path.join(sd, "timeline.txt")).read())
    pe = os.path.join(sd, "pmc_entry.json")
    if os.path.exists(pe) and os.path.getsize(pe) > 10:
        ent = json.load(open(pe))
        for k, v in ent.items():
            if v.get("bytes_per_step", 0) > 0:
                tab[k] = v
                print("pmc entry", k, v["bytes_per_step"], v["kernel_source_sha"])
# counters, rehearsal log, trajectory pin
for src, dst in ((RND + "_rebomos4m_pmc_sq_tcp_counters.json",) * 2, (RND + "_aeam1m_pmc_sq_tcp_counters.json",) * 2,
                 (RND + "_trajectory_pin.json",) * 2, (RND + "_rehearse.log", RND + "_rehearse_one_gpu.log")):
    if os.path.exists(os.path.join(G, src)) and os.path.getsize(os.path.join(G, src)) > 10:
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
json.dump(tab, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
# SQ / TCP / GRBM counters of the path kernels -> profiles/pmc_binding.json (bench.py: roofline.binding), stamped with the
# hash of the kernel sources of THIS tree (the collection ran on them: collect_rNN.sh counters, then this script, no edit between)
sys.path.insert(0, ROOT)
import bench  # noqa: E402
bf = os.path.join(P, "pmc_binding.json")
btab = json.load(open(bf)) if os.path.exists(bf) else {}
for src, key in ((RND + "_rebomos4m_pmc_sq_tcp_counters.json", "rebomos:24x24x24:1"), (RND + "_aeam1m_pmc_sq_tcp_counters.json", "aeam:63x63x63:1")):
    f = os.path.join(P, src)
    if os.path.exists(f) and os.path.getsize(f) > 10:
        btab[key] = {"kernel_source_sha": bench.kernel_source_sha(), "kernels": json.load(open(f)),
                     "note": "median dispatch per kernel and counter over the force-only steps of `bench.py --steps 4 --warmup 1`, "
                             "one counter group per rocprofv3 --pmc run (profiles/pmc_passes.sh, profiles/summarize_pmc.py); stored as profiles/" + src}
        print("binding entry", key, btab[key]["kernel_source_sha"], sorted(btab[key]["kernels"]))
json.dump(btab, open(bf, "w"), indent=1)
