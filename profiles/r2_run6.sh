#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1; do
  OUT=$R/gpurun_out/r2e6b/lds$v; mkdir -p $OUT
  export MDP_AEAM_LDS=$v
  run() { name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --workload aeam --temp 863 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
  run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT
  run grbm GRBM_GUI_ACTIVE
  cd $R && python3 profiles/summarize_pmc.py $OUT > $OUT/summary.json; cat $OUT/summary.json; cd /tmp
done
