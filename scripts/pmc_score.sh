#!/bin/bash
# PMC counters of the pose step's scoring kernel, the chain alone on the chip (2000 hypotheses x 25k points, 11 runs):
#   bash scripts/pmc_score.sh      output: gpurun_out/pmc_score.txt
R=$PWD; O=$R/gpurun_out/pmc_score
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/a -- python3 scripts/ransac_only.py > $O/a.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 scripts/ransac_only.py > $O/b.log 2>&1
python3 scripts/pmc_summary.py k_score $(find $O -name "*counter_collection.csv") > $O.txt
cat $O.txt
