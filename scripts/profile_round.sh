#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench, PMC passes over k_lk
# (one hardware block group per pass, 4 launches of 50k keypoints each), and the un-profiled bench line.
# usage: bash scripts/profile_round.sh <tag>      outputs under gpurun_out/prof_<tag>/
set -e
TAG=${1:-r1}
R=$PWD
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/repo_prof && cp -r $R /tmp/repo_prof && cd /tmp/repo_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu --no-frame-loop > $O/bench_under_rocprof.json 2> $O/trace.log
pmc() { n=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- python3 scripts/klt_only.py 50000 3 > $O/$n.log 2>&1
}
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/frame_trace -- python3 scripts/frame_only.py > $O/frame_loop.log 2> $O/frame_trace.log
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sqa SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pmc sqb SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
pmc tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
cd $R
cp $(find $O/trace -name "*kernel_stats.csv") $O/kernel_stats.csv
cp $(find $O/frame_trace -name "*kernel_stats.csv") $O/frame_loop_kernel_stats.csv
python3 scripts/pmc_summary.py k_lk $(find $O -name "*counter_collection.csv") > $O/k_lk_pmc.txt
python3 scripts/pmc_json.py 50000 4 $(find $O -name "*counter_collection.csv") > $O/k_lk_pmc.json
# the tracking launches of the profiled bench by grid size, and the 50k-keypoint launches by what ran beside them: alone
# on the chip (the roofline leg's laps, the stage_ms leg) or with the pose chain beside them (the timed steps: the kernel
# that waits for the head of the launch starts inside it)
python3 - "$(find $O/trace -name '*kernel_trace.csv' | head -1)" > $O/k_lk_by_grid.txt <<'PY'
import csv, sys, collections, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
waits = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_wait_head" in r["Kernel_Name"])
ends = [w[1] for w in waits]                # the waiting kernels do not overlap one another: sorted by start = sorted by end
d = collections.defaultdict(list)
for r in rows:
    if "k_lk" in r["Kernel_Name"]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        i = bisect.bisect_right(ends, s)    # the first waiting kernel that ends after this launch starts
        beside = i < len(waits) and waits[i][0] < e
        d[(int(r.get("Grid_Size", r.get("Grid_Size_X", 0))), beside)].append(e - s)
for (g, beside), v in sorted(d.items()):
    v.sort()
    print("k_lk grid %8d (%6d keypoints), %-28s: %5d launches, average %9.1f ns, median %9.1f ns, min %9.1f ns"
          % (g, g // 64, "pose chain beside it" if beside else "alone on the chip", len(v), sum(v) / len(v), v[len(v) // 2], v[0]))
PY
# the un-profiled bench line quotes traffic / valu_util from the PMC summary of THIS run (same box, same sources)
cp $O/k_lk_pmc.json $R/profiles/round3_k_lk_pmc.json
timeout -k 10 400 python3 bench.py --steps 50 --warmup 5 > $O/bench.json
cat $O/k_lk_pmc.txt; cat $O/k_lk_by_grid.txt; head -8 $O/kernel_stats.csv | cut -c1-160; cat $O/bench.json
