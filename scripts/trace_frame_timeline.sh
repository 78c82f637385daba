#!/bin/bash
# kernel timeline of two steady-state frames of the pipelined frame loop (start / duration / queue):
#   bash scripts/trace_frame_timeline.sh [frames in flight]
set -e
D=${1:-2}
R=$PWD; O=$R/gpurun_out/frame_timeline; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/repo_prof && cp -r $R /tmp/repo_prof && cd /tmp/repo_prof
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 scripts/frame_only.py 60 300 100 1 $D > $O/run.log 2> $O/trace.log
python3 - "$(find $O/t -name '*kernel_trace.csv' | head -1)" > $O/timeline.txt <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tr = [i for i, r in enumerate(rows) if "k_tracked" in r["Kernel_Name"]]
i0, i1 = tr[-12], tr[-10]                 # two frames in the middle of the last pass
t0 = int(rows[i0]["Start_Timestamp"])
per = [(int(rows[tr[k + 1]]["Start_Timestamp"]) - int(rows[tr[k]]["Start_Timestamp"])) / 1e3 for k in range(len(tr) - 30, len(tr) - 2)]
print("k_tracked to k_tracked, last frames (us):", " ".join("%.0f" % p for p in per))
qs = {}
for r in rows[i0 - 12:i1 + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Kernel_Name"])
    name = (m.group(1) if m else r["Kernel_Name"])[:24] + " g" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    q = r.get("Queue_Id", "?")
    col = qs.setdefault(q, len(qs))
    print("%s%-34s start %8.1f  dur %6.1f  end %8.1f" % ("    " * col, name, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
PY
tail -2 $O/run.log; cat $O/timeline.txt
