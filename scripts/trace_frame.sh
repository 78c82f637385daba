#!/bin/bash
# kernel trace of the frame loop at a given size: bash scripts/trace_frame.sh <landmarks> <new per frame>
set -e
R=$PWD; O=$R/gpurun_out/frame_trace_$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/repo_prof && cp -r $R /tmp/repo_prof && cd /tmp/repo_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 scripts/frame_only.py 24 $1 $2 > $O/run.log 2> $O/trace.log
cp $(find $O/t -name "*kernel_stats.csv") $O/kernel_stats.csv
tail -2 $O/run.log; head -14 $O/kernel_stats.csv | cut -c1-70,100-200
