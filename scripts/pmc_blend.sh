#!/bin/bash
# PMC counters for the compositing kernels (separate passes; no trace domains combined with --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "blend" --output-format csv -d $OUT/p1 -- python3 $R/scripts/kbench.py --reps 2 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-include-regex "blend" --output-format csv -d $OUT/p2 -- python3 $R/scripts/kbench.py --reps 2 > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_GDS --kernel-include-regex "blend" --output-format csv -d $OUT/p3 -- python3 $R/scripts/kbench.py --reps 2 > $OUT/p3.log 2>&1
find $OUT -name "*.csv" | head -20
tail -3 $OUT/p1.log
