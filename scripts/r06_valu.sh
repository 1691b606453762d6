#!/bin/bash
# Round 6: the VALU issue ceiling (scripts/dev/valu_ceiling.hip) by time, by s_memtime and under PMC counters, the per-tile
# timeline of the compositing kernels (scripts/dev/blend_timeline.py on a -DMTGS_TIMELINE variant) and the stall counters of the
# real kernels.  Inputs are prebuilt in the container: scripts/dev/_bin/valu_ceiling, mtgs_amd/_variants/libmtgs_rast_timeline.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_valu
rm -rf $OUT && mkdir -p $OUT
BIN=$R/scripts/dev/_bin/valu_ceiling
$BIN 4000 > $OUT/valu_time.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc1 -- $BIN 4000 > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU --output-format csv -d $OUT/pmc2 -- $BIN 4000 > $OUT/pmc2.log 2>&1
cd $R
timeout 600 python3 scripts/dev/blend_timeline.py mtgs_amd/_variants/libmtgs_rast_timeline.so 1920 1080 $OUT/timeline.npz > $OUT/timeline.txt 2>&1
timeout 600 python3 scripts/fbench.py > $OUT/fbench.txt 2>&1
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex "blend" --output-format csv -d $OUT/bp1 -- python3 $R/scripts/fbench.py --reps 3 > $OUT/bp1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM --kernel-include-regex "blend" --output-format csv -d $OUT/bp2 -- python3 $R/scripts/fbench.py --reps 3 > $OUT/bp2.log 2>&1
find $OUT -name "*.csv" | xargs ls -la
tail -30 $OUT/timeline.txt
