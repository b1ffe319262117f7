#!/bin/bash
# SQ / LDS / TCP counters of the AEAM kernels on the frozen configuration (profiles/aeam_frozen.py), one counter group
# per run (never combined with tracing).  usage: profiles/pmc_frozen.sh <outdir-under-gpurun_out>  (env selects variants)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/profiles/aeam_frozen.py 40 8 > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run grbm GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT && python3 profiles/summarize_pmc2.py $OUT "${PMC_KERNELS:-aeam_}"
rm -rf $OUT/*/*/*.db $OUT/*/*.db 2>/dev/null
