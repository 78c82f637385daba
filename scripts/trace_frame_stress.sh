#!/bin/bash
# kernel trace of the frame loop at the stress size, and the timeline of one steady-state frame:
#   bash scripts/trace_frame_stress.sh [frame_stress.py options]       outputs under gpurun_out/frame_stress/
set -e
R=$PWD; O=$R/gpurun_out/frame_stress; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/repo_prof && cp -r $R /tmp/repo_prof && cd /tmp/repo_prof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 scripts/frame_stress.py --passes 3 "$@" > $O/run.log 2> $O/trace.log
cp $(find $O/t -name "*kernel_stats.csv") $O/kernel_stats.csv
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" > $O/timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def name(r):
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
    return m.group(1) if m else r["Kernel_Name"][:20]
# the k_append_new launches end the frames: print everything between the 8th and the 10th from the end
ends = [i for i, r in enumerate(rows) if name(r) == "k_append_new"]
lo, hi = ends[-10], ends[-8]
t0 = int(rows[lo]["End_Timestamp"])
for r in rows[lo + 1: hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-16s q%-3s grid %-9s start %8.1f us  dur %7.1f us  end %8.1f" % (name(r)[:16], r["Queue_Id"], r["Grid_Size_X"], (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
PY
tail -3 $O/run.log; cat $O/timeline.txt
