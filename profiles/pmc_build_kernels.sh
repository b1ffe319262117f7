#!/bin/bash
# counters of the LIST-BUILD kernels (tile_scan, cand_build, tile_sort, classify, rev, tile_fill ...) of one forced
# reneighboring of the bench workload.  usage: profiles/pmc_build_kernels.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-host-mode $BENCH_ARGS > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
BENCH_ARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run grbm GRBM_GUI_ACTIVE
python3 - $OUT <<'PY'
import collections, csv, glob, json, os, re, sys
root = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", ""))
        if any(s in k for s in ("tile_scan", "cand_build", "cand_compact", "tile_sort", "tile_fill", "classify", "rev_kernel", "pack_cand", "tile_prune", "nbuild", "bin", "hold_all", "unit_")):
            res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in res.items():
    o = {c: max(v) for c, v in cs.items()}      # the largest dispatch = the full-size build
    cyc = o.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc > 0:
        o["ms_at_2p4GHz"] = cyc / 2.4e6
        o["valu_issue_frac"] = o.get("SQ_INSTS_VALU", 0) * 4.0 / (1024 * cyc)
        o["waves_waiting_share"] = o.get("SQ_WAIT_ANY", 0) / max(o.get("SQ_WAVE_CYCLES", 1), 1)
    out[k] = o
print(json.dumps(out, indent=1))
PY
