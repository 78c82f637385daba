#!/bin/bash
# LDS-side counters of the tracking kernel (one pass): array cycles, conflict cycles, unaligned stalls, instruction counts
# usage: bash scripts/pmc_lds.sh <tag> [lib]      output: gpurun_out/pmc_lds_<tag>.txt
TAG=${1:-x}; LIB=${2:-visual_odometry_amd/libvo_hip.so}
R=$PWD; O=$R/gpurun_out/pmc_lds_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VO_HIP_LIB=$R/$LIB
cd $R
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -- python3 scripts/klt_only.py 50000 3 > $O.log 2>&1
python3 scripts/pmc_summary.py k_lk $(find $O -name "*counter_collection.csv") > $O.txt
cat $O.txt
