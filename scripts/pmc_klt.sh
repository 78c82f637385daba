#!/bin/bash
# PMC passes over k_lk (scripts/klt_only.py 50000 3 = 4 launches); one hardware block group per pass.
set -e
R=$PWD
O=$R/gpurun_out/pmc_klt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cp -r $R /tmp/repo_pmc && cd /tmp/repo_pmc
run() { # name counters...
  n=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- python3 scripts/klt_only.py 50000 3 > $O/$n.log 2>&1
}
run a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM
cd $R
python3 scripts/pmc_summary.py k_lk $(find $O -name "*counter_collection.csv") > $O/summary.txt
cat $O/summary.txt
