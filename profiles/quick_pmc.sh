#!/bin/bash
# quick SQ counters for the default bench (scratch helper)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/quickpmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/sq1.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/sq2.err
cd $R; python3 profiles/summarize_pmc.py $OUT
